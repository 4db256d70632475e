"""efficientspeech_amd -- MI355X-native EfficientSpeech acoustic-model forward path.

Drop-in for the reference's `layers` package (layers/__init__.py:1 exports exactly these three
names); all arithmetic runs in hand-written HIP kernels behind the C-ABI of include/esmi.h.
"""
from .config import CONFIGS, ESConfig
from .networks import (Encoder, Fuse, AcousticDecoder, FeatureUpsampler, MelDecoder, PhonemeEncoder, Phoneme2Mel,
                       SelfAttention, MixFFN, get_mask_from_lengths)
from .model import EfficientSpeech, build_phoneme2mel, from_lightning_checkpoint, load_numpy_state_dict
from .scheduler import BucketedSynthesizer
from .hifigan import Generator as HifiGanGenerator, HifiGanConfig, HIFIGAN_CONFIGS, get_hifigan

__all__ = ["PhonemeEncoder", "MelDecoder", "Phoneme2Mel", "Encoder", "Fuse", "AcousticDecoder", "FeatureUpsampler",
           "SelfAttention", "MixFFN", "get_mask_from_lengths", "CONFIGS", "ESConfig", "build_phoneme2mel",
           "load_numpy_state_dict", "EfficientSpeech", "from_lightning_checkpoint", "BucketedSynthesizer", "HifiGanGenerator", "HifiGanConfig", "HIFIGAN_CONFIGS", "get_hifigan"]
