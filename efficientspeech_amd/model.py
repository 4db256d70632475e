"""Model-level wrapper: the arithmetic-free part of the reference's model.py.

`EfficientSpeech` mirrors the attribute names and call signatures demo.py / synthesize.py use on the reference's
LightningModule (`/root/reference/model.py:103-164`, `demo.py:66-67,122-124`, `synthesize.py:103-124`) without Lightning:
`.phoneme2mel` (the MI355X-native acoustic model of this package), `.hifigan` (whatever vocoder module the caller plugs
in -- the HiFi-GAN generator is SURVEY §8f-3, outside the path this repo accelerates), `forward(x)`, `predict_step(batch)`,
`load_from_checkpoint(...)` for Lightning-shaped checkpoint dicts.
"""
import json
import os

import torch
from torch import nn

from .config import CONFIGS, ESConfig, LJSPEECH_PITCH_STATS, LJSPEECH_ENERGY_STATS
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


def _stats_from(preprocess_config):
    """pitch / energy bin ranges: stats.json under preprocess_config['path']['preprocessed_path'] (model.py:127-130), the
    LJSpeech values shipped with the reference when no config / file is given."""
    if preprocess_config is not None:
        path = os.path.join(preprocess_config["path"]["preprocessed_path"], "stats.json")
        if os.path.exists(path):
            with open(path) as f:
                stats = json.load(f)
            return tuple(stats["pitch"][:2]), tuple(stats["energy"][:2])
    return LJSPEECH_PITCH_STATS, LJSPEECH_ENERGY_STATS


class EfficientSpeech(nn.Module):
    """Drop-in for the object demo.py / synthesize.py hold (`model.phoneme2mel`, `model.hifigan`, `model(x)`).

    Constructor arguments as `EfficientSpeech.__init__` (model.py:104-121); `lr` / `weight_decay` feed `make_train_step`,
    `max_epochs` and `wav_path` are accepted and ignored.  NOTE the reference's class default `decoder_kernel_size=3` differs from
    its CLI default 5 that every published checkpoint uses: like the reference, the class keeps 3, callers
    (and `from_config`) pass what their checkpoint was trained with.

    `hifigan`: a vocoder module mapping mel (B, 80, L) -> wav (B, 1, samples).  None (default): `predict_step` returns the
    channels-first mel in place of the waveform.  `efficientspeech_amd.get_hifigan(...)` / `HifiGanGenerator` is the HIP vocoder
    (the reference's `get_hifigan`, model.py:23-49); any module with that signature plugs in."""

    def __init__(self, preprocess_config=None, lr=1e-3, weight_decay=1e-6, max_epochs=5000, depth=2, n_blocks=2, block_depth=2,
                 reduction=4, head=1, embed_dim=128, kernel_size=3, decoder_kernel_size=3, expansion=1, wav_path="wavs",
                 hifigan_checkpoint=None, infer_device=None, verbose=False, hifigan=None):
        super().__init__()
        pitch_stats, energy_stats = _stats_from(preprocess_config)
        encoder = PhonemeEncoder(pitch_stats=pitch_stats, energy_stats=energy_stats, depth=depth, reduction=reduction, head=head,
                                 embed_dim=embed_dim, kernel_size=kernel_size, expansion=expansion)
        decoder = MelDecoder(dim=embed_dim // reduction, kernel_size=decoder_kernel_size, n_blocks=n_blocks,
                             block_depth=block_depth)
        self.phoneme2mel = Phoneme2Mel(encoder=encoder, decoder=decoder)
        # model.py:148: the reference always builds its vocoder from `hifigan_checkpoint`.  Here an explicitly plugged-in module wins;
        # else a checkpoint path that exists is loaded through the HIP generator; a path that does not exist (a Lightning checkpoint
        # written on another machine carries one in its hyper_parameters) leaves the slot empty -- `load_from_checkpoint` then
        # rebuilds the vocoder from the checkpoint's own `hifigan.*` weights.
        if hifigan is None and hifigan_checkpoint is not None:
            if os.path.exists(hifigan_checkpoint):
                from .hifigan import get_hifigan
                hifigan = get_hifigan(hifigan_checkpoint, infer_device=infer_device, verbose=verbose)
            else:
                import warnings
                warnings.warn(f"hifigan_checkpoint {hifigan_checkpoint!r} not found: no vocoder attached (predict_step returns the mel)")
        self.hifigan = hifigan
        self.hparams = dict(depth=depth, n_blocks=n_blocks, block_depth=block_depth, reduction=reduction, head=head,
                            embed_dim=embed_dim, kernel_size=kernel_size, decoder_kernel_size=decoder_kernel_size,
                            expansion=expansion, infer_device=infer_device, lr=lr, weight_decay=weight_decay)
        if infer_device is not None:
            self.to(infer_device)

    @classmethod
    def from_config(cls, name_or_cfg, **kw):
        """One of the three published sizes ('tiny' / 'small' / 'base') with the CLI hyper-parameters (decoder kernel 5)."""
        cfg = CONFIGS[name_or_cfg] if isinstance(name_or_cfg, str) else name_or_cfg
        return cls(depth=cfg.depth, n_blocks=cfg.n_blocks, block_depth=cfg.block_depth, reduction=cfg.reduction, head=cfg.head,
                   embed_dim=cfg.embed_dim, kernel_size=cfg.kernel_size, decoder_kernel_size=cfg.decoder_kernel_size,
                   expansion=cfg.expansion, **kw)

    def forward(self, x):
        """model.py:155-156: the training dict when `self.training`, else `predict_step(x)`.  With autograd enabled the training
        dict comes from the differentiable operator path (efficientspeech_amd/train.py), so `loss(...)[...].backward()` works as
        it does on the reference; under `torch.no_grad()` it is the inference kernels' teacher-forced forward."""
        if not self.training:
            return self.predict_step(x)
        if torch.is_grad_enabled():
            from . import train
            out = train.train_forward(self.phoneme2mel, x)        # (pads to max(x["mel_len"]) with one .item(), as networks.py:344)
            B, T = x["phoneme"].shape
            return {"mel": out["mel"], "pitch": out["pitch"].reshape(B, T, 1), "energy": out["energy"].reshape(B, T, 1),
                    "duration": out["duration"].reshape(B, T, 1), "mel_len": out["mel_len"]}
        return self.phoneme2mel(x, train=True)

    def loss(self, y_hat, y, x):
        """model.py:167-209: (mel_loss, pitch_loss, energy_loss, duration_loss) of a training dict.  The four values are for reporting;
        `training_step` returns the differentiable weighted total the fused loss kernel computed alongside them."""
        from . import train
        from .networks import _mask_u8
        B, T = x["phoneme"].shape
        f = lambda t: t.contiguous().float()      # noqa: E731
        parts, total = train._Loss.apply(y_hat["mel"], y_hat["pitch"].reshape(B, T), y_hat["energy"].reshape(B, T),
                                         y_hat["duration"].reshape(B, T), f(y["mel"]), f(x["pitch"]), f(x["energy"]),
                                         x["duration"].to(torch.int32).contiguous(), _mask_u8(x["mel_mask"]), _mask_u8(x["phoneme_mask"]))
        self._last_total = total                   # the differentiable weighted sum; the four parts are reported values
        return parts[0], parts[1], parts[2], parts[3]

    def training_step(self, batch, batch_idx=0):
        """model.py:212-226: the weighted total 10 mel + 2 pitch + 2 energy + duration (call `.backward()` on it)."""
        x, y = batch
        self.loss(self.forward(x), y, x)
        return self._last_total                    # the fused loss kernel's own total: its backward seeds all four terms

    def make_train_step(self, **kw):
        """`configure_optimizers` + the step Lightning would drive (model.py:279-283): an `efficientspeech_amd.train.TrainStep`
        over the acoustic model with this module's lr / weight_decay."""
        from . import train
        kw.setdefault("lr", self.hparams.get("lr", 1e-3))
        kw.setdefault("weight_decay", self.hparams.get("weight_decay", 1e-6))
        return train.TrainStep(self.phoneme2mel, **kw)

    def predict_step(self, batch, batch_idx=0, dataloader_idx=0):
        """model.py:159-164: (wav, mel_len, duration); wav = hifigan(mel.transpose(1, 2)).squeeze(1).  Without a vocoder the
        first element is the channels-first mel (B, 80, L) itself."""
        mel, mel_len, duration = self.phoneme2mel(batch, train=False)
        mel = mel.transpose(1, 2)
        if self.hifigan is None:
            return mel, mel_len, duration
        return self.hifigan(mel).squeeze(1), mel_len, duration

    @classmethod
    def load_from_checkpoint(cls, checkpoint, map_location=None, strict=True, **hparams):
        """`checkpoint`: path of a Lightning .ckpt (torch.load) or the already-loaded dict
        {'state_dict': {'phoneme2mel.*', 'hifigan.*'}, 'hyper_parameters': {...}}.  Constructor arguments come from
        `hparams`, falling back to the checkpoint's own `hyper_parameters` (as Lightning does, synthesize.py:103-119).
        `phoneme2mel.*` loads strict; `hifigan.*` goes to the attached vocoder, and when none is attached the HIP generator is built
        from those weights (v1 / v2 / v3 recognised by conv_pre's width), so `predict_step` returns waveforms as on the reference."""
        ckpt = torch.load(checkpoint, map_location=map_location or "cpu") if isinstance(checkpoint, (str, os.PathLike)) else checkpoint
        hp = dict(ckpt.get("hyper_parameters", {}))
        hp.update(hparams)
        accepted = cls.__init__.__code__.co_varnames[1:cls.__init__.__code__.co_argcount]
        model = cls(**{k: v for k, v in hp.items() if k in accepted})
        sd = ckpt["state_dict"]
        model.phoneme2mel.load_state_dict({k[len("phoneme2mel."):]: v for k, v in sd.items() if k.startswith("phoneme2mel.")},
                                          strict=strict)
        voc = {k[len("hifigan."):]: v for k, v in sd.items() if k.startswith("hifigan.")}
        if "conv_pre.weight" in voc and model.hifigan is None:
            # the checkpoint owns a vocoder (the reference's module does, model.py:148) and none is attached: rebuild the generator
            # from the weights themselves -- v1 / v2 / v3 differ in conv_pre's width (512 / 128 / 256 channels)
            from .hifigan import HIFIGAN_CONFIGS, Generator
            width = int(voc["conv_pre.weight"].shape[0])
            name = {512: "v1", 128: "v2", 256: "v3"}.get(width)
            if name is None:
                raise RuntimeError(f"hifigan.* weights of an unknown generator (conv_pre width {width})")
            model.hifigan = Generator(HIFIGAN_CONFIGS[name])
            dev = hp.get("infer_device")
            if dev is not None:
                model.hifigan.to(dev)
        if voc and hasattr(model.hifigan, "load_state_dict"):
            model.hifigan.load_state_dict(voc, strict=strict)
        return model.eval()
