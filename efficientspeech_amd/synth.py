"""Checkpoint key table + seeded synthetic weights for the acoustic model.

`state_dict_spec(cfg)` lists every tensor of `Phoneme2Mel.state_dict()` in the
reference's order with its shape and role (probed against
/root/reference/layers/networks.py by tools/gen_golden.py, which asserts name and
shape equality with the reference modules).

`synth_state_dict(cfg, seed)` fills those tensors from NumPy PCG64 streams keyed
by (seed, crc32(name)) so that the build container (where the reference is
importable) and the GPU box (where it is not) construct bit-identical weights
without shipping any checkpoint.  No trained checkpoint exists offline
 and random-init durations round to 0, so the duration /
pitch / energy heads get a bias that makes the synthetic model produce usable
durations (2..6 frames) and spread pitch/energy over several bins.
"""
import zlib
from collections import OrderedDict

import numpy as np

from .config import ESConfig, N_SYMBOLS, LJSPEECH_PITCH_STATS, LJSPEECH_ENERGY_STATS

# roles: how a tensor is initialised
MATRIX, LN_GAIN, BIAS, EMBED_PAD0, EMBED, BINS_PITCH, BINS_ENERGY = range(7)


def state_dict_spec(cfg: ESConfig):
    """[(key, shape, role)] in reference state_dict order (prefix-free: keys start at
    `encoder.` / `decoder.` exactly as under `phoneme2mel.` in a Lightning ckpt)."""
    spec = []
    add = lambda k, s, r: spec.append((k, tuple(int(v) for v in s), r))
    dim, E = cfg.dim, cfg.embed_dim
    # --- Encoder (networks.py:18-50)
    add("encoder.encoder.embed.weight", (N_SYMBOLS + 1, E), EMBED_PAD0)
    dim_ins = [E] + [dim * 2 ** i for i in range(cfg.depth - 1)]
    dim_outs = [dim * 2 ** i for i in range(cfg.depth)]
    for i, (ci, co) in enumerate(zip(dim_ins, dim_outs)):
        h = cfg.head * (i + 1)
        k = cfg.kernel_size - (2 if i > 0 else 0)
        p = f"encoder.encoder.attn_blocks.{i}."
        add(p + "0.weight", (ci, ci, k), MATRIX)
        add(p + "1.weight", (co, ci, 1), MATRIX)
        add(p + "2.qkv.weight", (3 * h * co, co), MATRIX)
        add(p + "2.proj.weight", (co, h * co), MATRIX)
        add(p + "2.proj.bias", (co,), BIAS)
        e = co * cfg.expansion
        add(p + "3.mlp1.weight", (e, co), MATRIX)
        add(p + "3.mlp1.bias", (e,), BIAS)
        add(p + "3.conv.weight", (e, e, 3), MATRIX)
        add(p + "3.conv.bias", (e,), BIAS)
        add(p + "3.mlp2.weight", (co, e), MATRIX)
        add(p + "3.mlp2.bias", (co,), BIAS)
        add(p + "4.weight", (co,), LN_GAIN)
        add(p + "4.bias", (co,), BIAS)
        add(p + "5.weight", (co,), LN_GAIN)
        add(p + "5.bias", (co,), BIAS)
    # --- Fuse (networks.py:171-187)
    for i, d in enumerate(dim_outs):
        p = f"encoder.fuse.mlps.{i}."
        add(p + "0.weight", (dim, d), MATRIX)
        add(p + "0.bias", (dim,), BIAS)
        if d // dim > 1:
            add(p + "1.weight", (dim, dim, cfg.kernel_size), MATRIX)   # ConvTranspose1d (Cin,Cout,k)
            add(p + "1.bias", (dim,), BIAS)
    add("encoder.fuse.fuse.weight", (dim, dim * cfg.depth), MATRIX)
    add("encoder.fuse.fuse.bias", (dim,), BIAS)
    # --- 3x AcousticDecoder (networks.py:93-125); registration order pitch, energy, duration
    for which in ("pitch", "energy", "duration"):
        p = f"encoder.{which}_decoder."
        if which == "pitch":
            add(p + "pitch_bins", (dim - 1,), BINS_PITCH)
        if which == "energy":
            add(p + "energy_bins", (dim - 1,), BINS_ENERGY)
        add(p + "conv1.0.weight", (dim, dim, 3), MATRIX)
        add(p + "conv1.0.bias", (dim,), BIAS)
        add(p + "norm1.weight", (dim,), LN_GAIN)
        add(p + "norm1.bias", (dim,), BIAS)
        add(p + "conv2.0.weight", (dim, dim, 3), MATRIX)
        add(p + "conv2.0.bias", (dim,), BIAS)
        add(p + "norm2.weight", (dim,), LN_GAIN)
        add(p + "norm2.bias", (dim,), BIAS)
        add(p + "linear.weight", (1, dim), MATRIX)
        add(p + "linear.bias", (1,), BIAS)
        if which in ("pitch", "energy"):
            add(p + f"{which}_embedding.weight", (dim, dim), EMBED)
    # --- MelDecoder (networks.py:264-288)
    dx2, d4, kd = cfg.dx2, cfg.d4, cfg.decoder_kernel_size
    add("decoder.proj.0.weight", (dx2, d4), MATRIX)
    add("decoder.proj.0.bias", (dx2,), BIAS)
    add("decoder.proj.2.weight", (dx2,), LN_GAIN)
    add("decoder.proj.2.bias", (dx2,), BIAS)
    for b in range(cfg.n_blocks):
        for d in range(cfg.block_depth):
            p = f"decoder.blocks.{b}.0.{d}."
            add(p + "0.0.weight", (dx2, 1, kd), MATRIX)     # depthwise
            add(p + "0.0.bias", (dx2,), BIAS)
            add(p + "0.1.weight", (dx2, dx2, 1), MATRIX)    # pointwise
            add(p + "0.1.bias", (dx2,), BIAS)
            add(p + "1.weight", (dx2,), LN_GAIN)
            add(p + "1.bias", (dx2,), BIAS)
        add(f"decoder.blocks.{b}.1.weight", (dx2,), LN_GAIN)
        add(f"decoder.blocks.{b}.1.bias", (dx2,), BIAS)
    add("decoder.mel_linear.weight", (cfg.n_mel_channels, dx2), MATRIX)
    add("decoder.mel_linear.bias", (cfg.n_mel_channels,), BIAS)
    return spec


def param_count(cfg: ESConfig) -> int:
    return sum(int(np.prod(s)) for _, s, _ in state_dict_spec(cfg))


def linspace_f32(lo, hi, n):
    """Bin edges exactly as the reference builds them: torch.linspace(lo, hi, n) in
    float32 (networks.py:109,118).  ATen's vectorised kernel is not reproducible with
    scalar NumPy arithmetic (checked: 6 of 31 edges differ by 1 ulp), so torch is called
    here; golden fixtures additionally store the edges they were generated with."""
    import torch
    return torch.linspace(lo, hi, n).numpy().copy()


def _rng(seed, key):
    return np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(key.encode())]))


def synth_state_dict(cfg: ESConfig, seed: int = 1234,
                     pitch_stats=LJSPEECH_PITCH_STATS, energy_stats=LJSPEECH_ENERGY_STATS):
    """OrderedDict key -> float32 ndarray, deterministic in (cfg, seed)."""
    sd = OrderedDict()
    for key, shape, role in state_dict_spec(cfg):
        g = _rng(seed, key)
        if role == MATRIX:
            fan_in = int(np.prod(shape[1:]))
            w = g.standard_normal(shape, dtype=np.float64) / np.sqrt(fan_in)
        elif role == LN_GAIN:
            w = 1.0 + 0.1 * g.standard_normal(shape, dtype=np.float64)
        elif role == BIAS:
            w = 0.1 * g.standard_normal(shape, dtype=np.float64)
        elif role == EMBED_PAD0:
            w = g.standard_normal(shape, dtype=np.float64)
            w[0] = 0.0                                  # padding_idx=0 row (networks.py:32)
        elif role == EMBED:
            w = g.standard_normal(shape, dtype=np.float64)
        elif role == BINS_PITCH:
            w = linspace_f32(pitch_stats[0], pitch_stats[1], shape[0])
        elif role == BINS_ENERGY:
            w = linspace_f32(energy_stats[0], energy_stats[1], shape[0])
        else:
            raise AssertionError(role)
        sd[key] = np.ascontiguousarray(w, dtype=np.float32)
    # make the heads usable with random weights
    sd["encoder.duration_decoder.linear.weight"] *= np.float32(2.0)
    sd["encoder.duration_decoder.linear.bias"][:] = np.float32(3.6)
    sd["encoder.pitch_decoder.linear.weight"] *= np.float32(4.0)
    sd["encoder.pitch_decoder.linear.bias"][:] = np.float32(4.0)
    sd["encoder.energy_decoder.linear.weight"] *= np.float32(3.0)
    sd["encoder.energy_decoder.linear.bias"][:] = np.float32(3.0)
    return sd


def synth_phonemes(B, T, seed=1234, lengths=None):
    """int32 (B,T) ids uniform in [1, N_SYMBOLS]; positions >= lengths[b] are PAD (0).
    Returns (phoneme int32, phoneme_mask bool) -- mask True = padding
    (utils/tools.py:43-51 get_mask_from_lengths convention)."""
    g = _rng(seed, f"phoneme/{B}x{T}")
    ids = g.integers(1, N_SYMBOLS + 1, size=(B, T), dtype=np.int64).astype(np.int32)
    if lengths is None:
        mask = np.zeros((B, T), dtype=bool)
    else:
        lengths = np.asarray(lengths)
        mask = np.arange(T)[None, :] >= lengths[:, None]
        ids[mask] = 0
    return ids, mask
