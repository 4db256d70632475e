#!/usr/bin/env python3
"""Development tool: time mel_decoder_kernel alone (fused-LR mode, D-const) for one or more library builds.
   python tools/bench_decoder.py [--config tiny] [--libs a.so b.so ...]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from efficientspeech_amd import CONFIGS, _lib, build_phoneme2mel, load_numpy_state_dict
from efficientspeech_amd.synth import synth_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="tiny"); ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--phonemes", type=int, default=128); ap.add_argument("--dur", type=int, default=6)
ap.add_argument("--zeros", action="store_true", help="zero activations (DVFS probe: same instruction stream, less switching power)"); ap.add_argument("--iters", type=int, default=30); ap.add_argument("--h0", action="store_true", help="phoneme-rate first stage supplied (what the full forward does for tiny, T <= 128)"); ap.add_argument("--burst", type=int, default=20, help="launches per timed burst (back-to-back, one event pair)"); ap.add_argument("--libs", nargs="*", default=[_lib.LIB_PATH])
a = ap.parse_args()
cfg = CONFIGS[a.config]
B, T, L = a.batch, a.phonemes, a.phonemes * a.dur
flops = {"tiny": 189_440, "small": 973_824, "base": 1_505_792}[a.config]
ref = None
for path in a.libs:
    _lib._LIB = _lib.bind(C.CDLL(os.path.abspath(path)))
    net = build_phoneme2mel(cfg); load_numpy_state_dict(net, synth_state_dict(cfg)); net = net.cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    feat = torch.randn((B, T, cfg.d4), device="cuda", generator=g)
    cum = (torch.arange(1, T + 1, device="cuda", dtype=torch.int32) * a.dur).repeat(B, 1).contiguous()
    mel_len = torch.full((B,), L, dtype=torch.int32, device="cuda")
    dec = net.decoder
    h0 = torch.randn((B, T, cfg.dx2), device="cuda", generator=g) if a.h0 else None
    if a.zeros:
        feat.zero_()
        if h0 is not None: h0.zero_()
    for _ in range(5):
        mel = dec._fused(feat, cum, mel_len, None, L, True, L, h0=h0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.burst):
            mel = dec._fused(feat, cum, mel_len, None, L, True, L, h0=h0)
        e.record()
        torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / a.burst)
    ms = float(np.median(ts))
    if ref is None: ref = mel.clone()
    diff = float((mel - ref).abs().max())
    print(f"{os.path.basename(path):32s} {ms*1e3:8.1f} us  min {min(ts)*1e3:8.1f}  {flops*B*L/ms/1e9:6.1f} TF  "
          f"{flops*B*L/ms/1e9/157.3*100:5.1f}% fp32 peak   max|diff vs first| {diff:.2e}", flush=True)
