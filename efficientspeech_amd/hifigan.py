"""HiFi-GAN generator (the vocoder `model.py:161-162` feeds the mel into) on MI355X -- SURVEY §8f-3.

Host-side mirror of `/root/reference/hifigan/models.py:84-135` (`Generator`, `ResBlock1` :20-58, `ResBlock2` :61-82): same
attribute names (`conv_pre`, `ups`, `resblocks[*].convs1/convs2`, `conv_post`), so the `hifigan.*` tensors of a Lightning
checkpoint (saved AFTER `remove_weight_norm()`, model.py:44: `weight` / `bias` per conv) load with strict=True; the stand-alone
generator files of the reference (`hifigan/LJ_V2/generator_v2`: `{"generator": {... weight_g, weight_v ...}}`) are folded
by `fold_weight_norm()`.  The arithmetic is one C-ABI call, `esmi_hifigan_generator_f32` (include/esmi.h).

Layout: the reference takes the mel channels-first (B, 80, L) -- `model.py:161` transposes the acoustic model's (B, L, 80)
output for it.  The kernels are channels-last throughout, so `forward` undoes that transpose as a free view when it is given the
transposed tensor (the `EfficientSpeech.predict_step` hand-off): no copy, no channels-first epilogue needed.
"""
import ctypes as C
import json
import os
import zlib
from collections import OrderedDict
from dataclasses import dataclass, field

import numpy as np
import torch
from torch import nn

from . import _lib
from .networks import _PackCache, _PackedModule, _f32, _on_device_of, _ptr, _check_split_range
from . import networks

LRELU_SLOPE = 0.1      # hifigan/models.py:17


@dataclass(frozen=True)
class HifiGanConfig:
    """The fields of hifigan/*/config.json that define the generator (defaults = LJ_V2, the reference's default vocoder)."""
    resblock: str = "1"
    upsample_rates: tuple = (8, 8, 2, 2)
    upsample_kernel_sizes: tuple = (16, 16, 4, 4)
    upsample_initial_channel: int = 128
    resblock_kernel_sizes: tuple = (3, 7, 11)
    resblock_dilation_sizes: tuple = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    num_mels: int = 80

    @classmethod
    def from_json(cls, path_or_dict):
        d = json.load(open(path_or_dict)) if isinstance(path_or_dict, (str, os.PathLike)) else dict(path_or_dict)
        return cls(resblock=str(d["resblock"]), upsample_rates=tuple(d["upsample_rates"]),
                   upsample_kernel_sizes=tuple(d["upsample_kernel_sizes"]),
                   upsample_initial_channel=int(d["upsample_initial_channel"]),
                   resblock_kernel_sizes=tuple(d["resblock_kernel_sizes"]),
                   resblock_dilation_sizes=tuple(tuple(x) for x in d["resblock_dilation_sizes"]),
                   num_mels=int(d.get("num_mels", 80)))

    @property
    def hop(self):
        return int(np.prod(self.upsample_rates))


HIFIGAN_CONFIGS = {
    "v1": HifiGanConfig(upsample_initial_channel=512),                                  # hifigan/LJ/config.json family
    "v2": HifiGanConfig(),                                                              # hifigan/LJ_V2/config.json
    "v3": HifiGanConfig(resblock="2", upsample_rates=(8, 8, 4), upsample_kernel_sizes=(16, 16, 8), upsample_initial_channel=256,
                        resblock_kernel_sizes=(3, 5, 7), resblock_dilation_sizes=((1, 2), (2, 6), (3, 12))),   # hifigan/LJ_V3
}


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


class ResBlock1(nn.Module):
    """Parameters of hifigan/models.py:20-46 (weight norm already removed)."""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.convs1 = nn.ModuleList([nn.Conv1d(channels, channels, kernel_size, 1, dilation=d, padding=get_padding(kernel_size, d))
                                     for d in dilation])
        self.convs2 = nn.ModuleList([nn.Conv1d(channels, channels, kernel_size, 1, dilation=1, padding=get_padding(kernel_size, 1))
                                     for _ in dilation])


class ResBlock2(nn.Module):
    """Parameters of hifigan/models.py:61-72."""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv1d(channels, channels, kernel_size, 1, dilation=d, padding=get_padding(kernel_size, d))
                                    for d in dilation])


class Generator(_PackedModule):
    """hifigan/models.py:84-135.  `h`: a HifiGanConfig, a config.json path / dict, or anything with the same attributes."""

    def __init__(self, h=None):
        super().__init__()
        if h is None:
            h = HifiGanConfig()
        elif isinstance(h, (str, os.PathLike, dict)) and not hasattr(h, "upsample_rates"):
            h = HifiGanConfig.from_json(h)
        elif not isinstance(h, HifiGanConfig):
            h = HifiGanConfig(resblock=str(h.resblock), upsample_rates=tuple(h.upsample_rates),
                              upsample_kernel_sizes=tuple(h.upsample_kernel_sizes),
                              upsample_initial_channel=int(h.upsample_initial_channel),
                              resblock_kernel_sizes=tuple(h.resblock_kernel_sizes),
                              resblock_dilation_sizes=tuple(tuple(x) for x in h.resblock_dilation_sizes))
        self.h = h
        self.num_kernels = len(h.resblock_kernel_sizes)
        self.num_upsamples = len(h.upsample_rates)
        c0 = h.upsample_initial_channel
        self.conv_pre = nn.Conv1d(h.num_mels, c0, 7, 1, padding=3)
        self.ups = nn.ModuleList([nn.ConvTranspose1d(c0 // (2 ** i), c0 // (2 ** (i + 1)), k, u, padding=(k - u) // 2)
                                  for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes))])
        rb = ResBlock1 if h.resblock == "1" else ResBlock2
        self.resblocks = nn.ModuleList()
        for i in range(len(self.ups)):
            ch = c0 // (2 ** (i + 1))
            for k, d in zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes):
                self.resblocks.append(rb(h, ch, k, d))
        self.conv_post = nn.Conv1d(ch, 1, 7, 1, padding=3)
        self._cache = _PackCache()
        self.fuse_resblocks = True    # one launch per ResBlock where the library has the kernel (False: one per convolution)
        for p in self.parameters():
            p.requires_grad = False

    def remove_weight_norm(self):
        """No-op: this mirror holds plain `weight` tensors (what the reference's module holds after model.py:44)."""

    # ------------------------------------------------------------------ packing
    def _packed(self, lib, stream):
        def build():
            h = self.h
            w = _lib.HifiGanWeights()
            s = _lib.HifiGanShape()
            keep = []

            def conv(m, transposed=False):
                wt, _ = networks._pack_conv(lib, stream, m.weight, transposed=transposed)     # tap-major (k, Cout, Cin)
                b = _f32(m.bias)
                keep.extend([wt, b])
                return _ptr(wt), _ptr(b), wt

            def frag(wt):
                """Fragment-ordered split-f16 planes for the one-launch ResBlock kernel (None: no such kernel for this shape)."""
                k, c, _ = wt.shape
                nbytes = lib.esmi_pack_resblock_bytes(c, k) if self.fuse_resblocks else 0
                if not nbytes:
                    return None
                dst = torch.empty(nbytes, dtype=torch.uint8, device=wt.device)
                lib.esmi_pack_resblock_f16(_ptr(wt), _ptr(dst), c, k, stream)
                keep.append(dst)
                return _ptr(dst)
            mats = []
            w.pre_w, w.pre_b, t = conv(self.conv_pre); mats.append(t)
            for i, up in enumerate(self.ups):
                w.up_w[i], w.up_b[i], t = conv(up, transposed=True); mats.append(t)
            nconv = 3 if h.resblock == "1" else 2
            for n, rb in enumerate(self.resblocks):
                for m in range(nconv):
                    if h.resblock == "1":
                        w.rb_w1[n * 3 + m], w.rb_b1[n * 3 + m], t = conv(rb.convs1[m]); mats.append(t)
                        w.rb_wp1[n * 3 + m] = frag(t)
                        w.rb_w2[n * 3 + m], w.rb_b2[n * 3 + m], t = conv(rb.convs2[m]); mats.append(t)
                        w.rb_wp2[n * 3 + m] = frag(t)
                    else:
                        w.rb_w1[n * 3 + m], w.rb_b1[n * 3 + m], t = conv(rb.convs[m]); mats.append(t)
                        w.rb_wp1[n * 3 + m] = frag(t)
            w.post_w, w.post_b, t = conv(self.conv_post); mats.append(t)
            _check_split_range(lib, stream, mats, "HiFi-GAN generator weights")
            s.n_mel, s.initial_channel, s.n_up, s.n_kernels = h.num_mels, h.upsample_initial_channel, self.num_upsamples, self.num_kernels
            s.resblock = 1 if h.resblock == "1" else 2
            for i in range(self.num_upsamples):
                s.up_rates[i], s.up_kernels[i] = h.upsample_rates[i], h.upsample_kernel_sizes[i]
            for j in range(self.num_kernels):
                s.rb_kernels[j] = h.resblock_kernel_sizes[j]
                for m, d in enumerate(h.resblock_dilation_sizes[j]):
                    s.rb_dilations[j * 3 + m] = d
            return w, s, keep
        return self._cache.get(lambda: list(self.parameters()), build)

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        """x: mel (B, num_mels, L) as the reference takes it -> wav (B, 1, L * hop)."""
        with _on_device_of(self.conv_post.weight):
            return self._forward(x)

    def _forward(self, x):
        lib, stream = networks._runtime(self.conv_post.weight)
        mel = _f32(x.transpose(1, 2))                  # (B, L, 80) channels-last: a free view of the acoustic model's own output
        B, L, nm = mel.shape
        assert nm == self.h.num_mels, f"expected {self.h.num_mels} mel channels, got {nm}"
        wav = torch.empty((B, 1, L * self.h.hop), dtype=torch.float32, device=mel.device)
        if B == 0 or L == 0:
            return wav
        w, s, _keep = self._packed(lib, stream)
        nbytes = lib.esmi_hifigan_workspace_bytes(C.byref(s), B, L)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=mel.device)
        lib.esmi_hifigan_generator_f32(C.byref(w), C.byref(s), _ptr(mel), B, L, _ptr(wav), _ptr(ws), nbytes, stream)
        return wav


# ---------------------------------------------------------------------- checkpoints
def fold_weight_norm(sd):
    """{... 'x.weight_g', 'x.weight_v' ...} (torch.nn.utils.weight_norm, dim 0) -> {... 'x.weight' ...}: w = g * v / ||v||,
    the norm over all dims but the first -- what remove_weight_norm() leaves in `weight` (hifigan/models.py:125-133)."""
    out = OrderedDict()
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            base = k[:-len(".weight_g")]
            g, vv = v.float(), sd[base + ".weight_v"].float()
            norm = vv.reshape(vv.shape[0], -1).norm(dim=1).reshape(g.shape)
            out[base + ".weight"] = vv * (g / norm)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = v
    return out


def get_hifigan(checkpoint="hifigan/LJ_V2/generator_v2", infer_device=None, verbose=False):
    """model.py:23-49: config.json next to the checkpoint, `ckpt['generator']` (weight-norm form), eval, on `infer_device`."""
    main_path = os.path.dirname(os.path.abspath(checkpoint))
    cfg = HifiGanConfig.from_json(os.path.join(main_path, "config.json"))
    if verbose:
        print("Using hifigan checkpoint: ", checkpoint)
    ckpt = torch.load(checkpoint, map_location="cpu")
    voc = Generator(cfg)
    voc.load_state_dict(fold_weight_norm(ckpt["generator"]), strict=True)
    voc.eval()
    if infer_device is not None:
        voc.to(infer_device)
    return voc


def flops_per_mel_frame(h: HifiGanConfig):
    """Algorithmic FLOPs (2 per multiply-add, convolutions only) the generator spends per input mel frame."""
    c, rate, f = h.upsample_initial_channel, 1, 2 * h.num_mels * h.upsample_initial_channel * 7
    for u, k in zip(h.upsample_rates, h.upsample_kernel_sizes):
        rate *= u
        f += rate * 2 * c * (c // 2) * (k // u)
        c //= 2
        per_conv = 2 if h.resblock == "1" else 1
        for kk, d in zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes):
            f += rate * len(d) * per_conv * 2 * c * c * kk
    return f + rate * 2 * c * 7


def hifigan_state_dict_spec(h: HifiGanConfig):
    """[(key, shape)] of Generator.state_dict() after remove_weight_norm, in the reference's registration order."""
    spec = []
    c0 = h.upsample_initial_channel
    spec += [("conv_pre.weight", (c0, h.num_mels, 7)), ("conv_pre.bias", (c0,))]
    for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
        spec += [(f"ups.{i}.weight", (c0 // 2 ** i, c0 // 2 ** (i + 1), k)), (f"ups.{i}.bias", (c0 // 2 ** (i + 1),))]
    n = 0
    for i in range(len(h.upsample_rates)):
        ch = c0 // 2 ** (i + 1)
        for k, d in zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes):
            names = ("convs1", "convs2") if h.resblock == "1" else ("convs",)
            for nm in names:
                for m in range(len(d)):
                    spec += [(f"resblocks.{n}.{nm}.{m}.weight", (ch, ch, k)), (f"resblocks.{n}.{nm}.{m}.bias", (ch,))]
            n += 1
    spec += [("conv_post.weight", (1, ch, 7)), ("conv_post.bias", (1,))]
    return spec


def synth_hifigan_state_dict(h: HifiGanConfig, seed=1234, gain=0.7):
    """Seeded synthetic generator weights by key name (NumPy PCG64, as synth.synth_state_dict): no trained vocoder travels to
    the GPU box.  Matrices ~ N(0, gain / sqrt(fan_in)) keep the ~50-layer chain O(1) up to the final tanh."""
    sd = OrderedDict()
    for key, shape in hifigan_state_dict_spec(h):
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))
        if key.endswith(".bias"):
            sd[key] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
        else:
            if key.startswith("ups."):
                fan_in = shape[0] * shape[2] / max(1, h.upsample_rates[int(key.split(".")[1])])   # taps that reach one output
            else:
                fan_in = shape[1] * shape[2]
            sd[key] = (gain / np.sqrt(fan_in) * rng.standard_normal(shape)).astype(np.float32)
    return sd
