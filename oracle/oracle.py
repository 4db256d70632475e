"""ctypes front-end of the CPU oracle (oracle/es_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py; the product package never imports this module.
All arrays are NumPy, channels-last, float32 / int32 / bool.
"""
import ctypes as C
import os
import subprocess
from types import SimpleNamespace

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


class _Cfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("depth", "reduction", "head", "embed_dim", "kernel_size", "expansion",
                                       "n_blocks", "block_depth", "dec_kernel", "n_mel", "vocab")]


class _Weights(C.Structure):
    _fields_ = [("n", C.c_int), ("names", C.POINTER(C.c_char_p)), ("ptrs", C.POINTER(C.c_void_p))]


def build(force=False):
    """Compile both oracle variants with gcc (called by __graft_entry__.build())."""
    src = os.path.join(_HERE, "es_oracle.c")
    libs = [os.path.join(_HERE, f) for f in ("libes_oracle.so", "libes_oracle_f32.so")]
    if force or not all(os.path.exists(f) and os.path.getmtime(f) >= os.path.getmtime(src) for f in libs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def lib(f32=False):
    name = "libes_oracle_f32.so" if f32 else "libes_oracle.so"
    if name not in _LIBS:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.eso_block_len.restype = C.c_int
        L.eso_acc_bytes.restype = C.c_int
        for fn in ("eso_phoneme_encoder", "eso_mel_decoder", "eso_encoder", "eso_fuse", "eso_acoustic", "eso_self_attention",
                   "eso_mixffn"):
            getattr(L, fn).restype = C.c_int
        _LIBS[name] = L
    return _LIBS[name]


def _cfg(cfg):
    from efficientspeech_amd.config import N_SYMBOLS
    return _Cfg(cfg.depth, cfg.reduction, cfg.head, cfg.embed_dim, cfg.kernel_size, cfg.expansion,
                cfg.n_blocks, cfg.block_depth, cfg.decoder_kernel_size, cfg.n_mel_channels, N_SYMBOLS + 1)


class Weights:
    """Keeps the name/pointer tables (and the arrays behind them) alive."""

    def __init__(self, sd):
        self.arrays = {k: np.ascontiguousarray(np.asarray(v), dtype=np.float32) for k, v in sd.items()}
        keys = list(self.arrays)
        self._names = (C.c_char_p * len(keys))(*[k.encode() for k in keys])
        self._ptrs = (C.c_void_p * len(keys))(*[self.arrays[k].ctypes.data for k in keys])
        self.c = _Weights(len(keys), self._names, self._ptrs)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"es_oracle {what} failed with code {rc}")


def block_len(cfg, T, i):
    c = _cfg(cfg)
    return lib().eso_block_len(C.byref(c), int(T), int(i))


def phoneme_encoder(cfg, w: Weights, phoneme, mask=None, pitch=None, energy=None, duration=None, f32=False, taps=False):
    """PhonemeEncoder.forward up to the rounded durations (networks.py:336-384).
    mask: bool (B,T), required iff B>1.  pitch/energy/duration: teacher values (train=True) or None."""
    L = lib(f32)
    phoneme = np.ascontiguousarray(phoneme, dtype=np.int32)
    B, T = phoneme.shape
    dim = cfg.dim
    m8 = None if mask is None else np.ascontiguousarray(mask).astype(np.uint8)
    pt = None if pitch is None else np.ascontiguousarray(pitch, dtype=np.float32)
    et = None if energy is None else np.ascontiguousarray(energy, dtype=np.float32)
    dt = None if duration is None else np.ascontiguousarray(duration, dtype=np.int32)
    o = SimpleNamespace(
        pitch=np.empty((B, T, 1), np.float32), energy=np.empty((B, T, 1), np.float32),
        duration=np.empty((B, T, 1), np.float32), pitch_idx=np.empty((B, T), np.int32),
        energy_idx=np.empty((B, T), np.int32), feat=np.empty((B, T, 4 * dim), np.float32),
        dur=np.empty((B, T), np.int32), mel_len=np.empty((B,), np.int32), f_taps=None, fused=None)
    tp, fz = None, None
    if taps:
        o.f_taps = [np.empty((B, block_len(cfg, T, i), dim << i), np.float32) for i in range(cfg.depth)]
        o.fused = np.empty((B, T, dim), np.float32)
        tp = (C.c_void_p * cfg.depth)(*[a.ctypes.data for a in o.f_taps])
        fz = _p(o.fused)
    c = _cfg(cfg)
    rc = L.eso_phoneme_encoder(C.byref(c), C.byref(w.c), B, T, _p(phoneme), _p(m8), _p(pt), _p(et), _p(dt),
                               _p(o.pitch), _p(o.energy), _p(o.duration), _p(o.pitch_idx), _p(o.energy_idx),
                               _p(o.feat), _p(o.dur), _p(o.mel_len), tp, fz)
    _chk(rc, "phoneme_encoder")
    o.mask = m8 if B > 1 else None
    return o


def length_regulate(dur, L):
    dur = np.ascontiguousarray(dur, dtype=np.int32)
    B, T = dur.shape
    idx = np.empty((B, L), np.int32)
    lib().eso_length_regulate(B, T, _p(dur), int(L), _p(idx))
    return idx


def upsample(feat, fmask, idx):
    feat = np.ascontiguousarray(feat, dtype=np.float32)
    B, T, Cc = feat.shape
    L = idx.shape[1]
    out = np.empty((B, L, Cc), np.float32)
    om = np.empty((B, L), np.uint8)
    fm = None if fmask is None else np.ascontiguousarray(fmask).astype(np.uint8)
    lib().eso_upsample(B, T, Cc, L, _p(feat), _p(fm), _p(np.ascontiguousarray(idx, dtype=np.int32)), _p(out), _p(om))
    return out, om.astype(bool)


def mel_decoder(cfg, w: Weights, features, f32=False):
    features = np.ascontiguousarray(features, dtype=np.float32)
    B, L, d4 = features.shape
    assert d4 == cfg.d4
    mel = np.empty((B, L, cfg.n_mel_channels), np.float32)
    c = _cfg(cfg)
    _chk(lib(f32).eso_mel_decoder(C.byref(c), C.byref(w.c), B, L, _p(features), _p(mel)), "mel_decoder")
    return mel


def phoneme2mel(cfg, w: Weights, phoneme, mask=None, pitch=None, energy=None, duration=None, max_mel_len=None,
                f32=False, taps=False):
    """Phoneme2Mel.forward (networks.py:415-434).  Teacher values given => the train=True data
    flow (targets bucketised / forced durations, padded to max_mel_len); else the eval flow.
    Returns a namespace with mel (B,L,80), mel_len, duration pred, and all intermediates."""
    o = phoneme_encoder(cfg, w, phoneme, mask, pitch, energy, duration, f32=f32, taps=taps)
    B = o.dur.shape[0]
    L = int(max_mel_len) if max_mel_len is not None else int(o.mel_len.max())
    o.idx = length_regulate(o.dur, L)
    o.features, o.masks = upsample(o.feat, o.mask, o.idx)
    o.mel = mel_decoder(cfg, w, o.features, f32=f32)
    if o.mask is not None and B > 1:
        o.mel[o.masks] = 0.0                       # networks.py:424-427
    else:
        o.masks = None
    return o


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def self_attention(x, qkv_w, proj_w, proj_b, heads):
    """SelfAttention.forward's tensor output (blocks.py:43-71; scores not masked); weights in checkpoint layout."""
    x, qkv_w, proj_w, proj_b = _f(x), _f(qkv_w), _f(proj_w), _f(proj_b)
    B, N, Cc = x.shape
    y = np.empty_like(x)
    _chk(lib().eso_self_attention(B, N, Cc, int(heads), _p(x), _p(qkv_w), _p(proj_w), _p(proj_b), _p(y)), "self_attention")
    return y


def mixffn(x, mlp1_w, mlp1_b, conv_w, conv_b, mlp2_w, mlp2_b):
    """MixFFN.forward (blocks.py:22-29); conv_w (eC, eC, 3) as stored in the checkpoint."""
    x = _f(x)
    B, N, Cc = x.shape
    e = mlp1_w.shape[0] // Cc
    y = np.empty_like(x)
    _chk(lib().eso_mixffn(B, N, Cc, int(e), _p(x), _p(_f(mlp1_w)), _p(_f(mlp1_b)), _p(_f(conv_w)), _p(_f(conv_b)),
                          _p(_f(mlp2_w)), _p(_f(mlp2_b)), _p(y)), "mixffn")
    return y


def acoustic(cfg, w: Weights, which, fused):
    """AcousticDecoder.forward (networks.py:151-165) of the pitch (0) / energy (1) / duration (2) predictor:
    -> (pred (B,T,1), features (B,T,dim) [LN2 output; the reference returns it for the duration predictor only])."""
    fused = _f(fused)
    B, T, dim = fused.shape
    pred = np.empty((B, T, 1), np.float32)
    feats = np.empty((B, T, dim), np.float32)
    c = _cfg(cfg)
    _chk(lib().eso_acoustic(C.byref(c), C.byref(w.c), int(which), B, T, _p(fused), _p(pred), _p(feats)), "acoustic")
    return pred, feats


class _HgCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n_mel", "initial_channel", "n_up", "n_kernels", "resblock")] + \
               [("up_rates", C.c_int * 8), ("up_kernels", C.c_int * 8), ("rb_kernels", C.c_int * 8),
                ("rb_dilations", (C.c_int * 3) * 8)]


def hifigan(h, w: Weights, mel):
    """HiFi-GAN Generator.forward (hifigan/models.py:84-135) on a channels-last mel (B, L, n_mel) -> wav (B, L * hop).
    `h`: efficientspeech_amd.hifigan.HifiGanConfig; `w`: Weights over the remove_weight_norm()-form state dict."""
    mel = _f(mel)
    B, L, nm = mel.shape
    c = _HgCfg()
    c.n_mel, c.initial_channel, c.n_up, c.n_kernels = nm, h.upsample_initial_channel, len(h.upsample_rates), len(h.resblock_kernel_sizes)
    c.resblock = 1 if str(h.resblock) == "1" else 2
    for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
        c.up_rates[i], c.up_kernels[i] = u, k
    for j, (k, d) in enumerate(zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes)):
        c.rb_kernels[j] = k
        for m, dd in enumerate(d):
            c.rb_dilations[j][m] = dd
    wav = np.empty((B, L * int(np.prod(h.upsample_rates))), np.float32)
    L_ = lib()
    L_.eso_hifigan.restype = C.c_int
    _chk(L_.eso_hifigan(C.byref(c), C.byref(w.c), B, L, _p(mel), _p(wav)), "hifigan")
    return wav


def mask_from_lengths(lengths, T):
    lengths = np.ascontiguousarray(lengths, dtype=np.int32)
    m = np.empty((len(lengths), T), np.uint8)
    lib().eso_mask_from_lengths(len(lengths), int(T), _p(lengths), _p(m))
    return m.astype(bool)


# ---------------------------------------------------------------------- training step: loss and optimizer
def training_loss(mel_pred, mel, mel_mask, pitch_pred, pitch, energy_pred, energy, dur_pred, dur, ph_mask):
    """model.py:167-216 in float64 NumPy: masked L1 on mel, masked MSE on pitch / energy / log(duration + 1) (means over the
    unmasked elements), total = 10 a + 2 b + 2 c + d.  Masks: True = padding.  Returns (four losses + total, gradients of the
    total with respect to mel_pred / pitch_pred / energy_pred / dur_pred)."""
    f = lambda a: np.asarray(a, np.float64)     # noqa: E731
    mel_pred, mel, pitch_pred, pitch, energy_pred, energy, dur_pred = map(f, (mel_pred, mel, pitch_pred, pitch, energy_pred, energy, dur_pred))
    mv = ~np.asarray(mel_mask, bool) if mel_mask is not None else np.ones(mel.shape[:2], bool)
    pv = ~np.asarray(ph_mask, bool) if ph_mask is not None else np.ones(pitch.shape, bool)
    n_el, n_ph = mv.sum() * mel.shape[-1], pv.sum()
    d = (mel_pred - mel) * mv[..., None]
    losses = [np.abs(d).sum() / n_el]
    grads = [10.0 * np.sign(d) / n_el]
    for pred, tgt, wgt in ((pitch_pred, pitch, 2.0), (energy_pred, energy, 2.0)):
        e = (pred.reshape(tgt.shape) - tgt) * pv
        losses.append((e * e).sum() / n_ph)
        grads.append(wgt * 2.0 * e / n_ph)
    dp = dur_pred.reshape(pitch.shape)
    e = (np.log(dp + 1.0) - np.log(f(dur) + 1.0)) * pv
    losses.append((e * e).sum() / n_ph)
    grads.append(2.0 * e / (dp + 1.0) / n_ph)
    losses.append(10.0 * losses[0] + 2.0 * losses[1] + 2.0 * losses[2] + losses[3])
    return np.array(losses), grads


def adamw_step(p, g, m, v, t, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-6):
    """torch.optim.AdamW's update (model.py:279-283 with its defaults), float32 arithmetic in the same order; t >= 1."""
    f32 = np.float32
    p, g, m, v = (np.asarray(a, f32) for a in (p, g, m, v))
    p = p * f32(1.0 - lr * weight_decay)
    m = f32(betas[0]) * m + f32(1.0 - betas[0]) * g
    v = f32(betas[1]) * v + f32(1.0 - betas[1]) * g * g
    bc1, bc2 = 1.0 - betas[0] ** t, 1.0 - betas[1] ** t
    denom = np.sqrt(v) / f32(np.sqrt(bc2)) + f32(eps)
    return (p - f32(lr / bc1) * (m / denom)).astype(f32), m.astype(f32), v.astype(f32)
