"""Model construction helpers (the arithmetic-free part of the reference's model.py:127-147)."""
import torch

from .config import ESConfig, LJSPEECH_PITCH_STATS, LJSPEECH_ENERGY_STATS
from .networks import PhonemeEncoder, MelDecoder, Phoneme2Mel


def build_phoneme2mel(cfg: ESConfig, pitch_stats=LJSPEECH_PITCH_STATS, energy_stats=LJSPEECH_ENERGY_STATS):
    """Same wiring as EfficientSpeech.__init__ (model.py:132-147): encoder + decoder -> Phoneme2Mel."""
    enc = PhonemeEncoder(pitch_stats=pitch_stats, energy_stats=energy_stats, **cfg.encoder_kwargs())
    dec = MelDecoder(**cfg.decoder_kwargs())
    return Phoneme2Mel(enc, dec).eval()


def load_numpy_state_dict(module, sd, strict=True):
    """Load {key: ndarray} (e.g. synth.synth_state_dict or the `phoneme2mel.*` slice of a Lightning ckpt)."""
    return module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=strict)


def from_lightning_checkpoint(ckpt, cfg: ESConfig, **kw):
    """Build Phoneme2Mel from a Lightning-shaped dict {'state_dict': {'phoneme2mel.*': ..., 'hifigan.*': ...}}
    (what demo.py:122 / synthesize.py:103-119 load); vocoder keys are ignored here."""
    net = build_phoneme2mel(cfg, **kw)
    sd = {k[len("phoneme2mel."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("phoneme2mel.")}
    net.load_state_dict(sd, strict=True)
    return net
