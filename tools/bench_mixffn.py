#!/usr/bin/env python3
"""Development: time MixFFN.forward (mlp1 -> dense k3 conv + GELU -> mlp2 through the LDS-staged GEMM kernel) at base ES block-1
size for one or more library builds.   python tools/bench_mixffn.py [--C 256] [--E 2] [--N 128] [--B 512] [--libs a.so b.so]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from efficientspeech_amd import _lib
from efficientspeech_amd.networks import MixFFN

ap = argparse.ArgumentParser()
ap.add_argument("--C", type=int, default=256); ap.add_argument("--E", type=int, default=2); ap.add_argument("--N", type=int, default=128)
ap.add_argument("--B", type=int, default=512); ap.add_argument("--iters", type=int, default=20); ap.add_argument("--libs", nargs="*", default=[_lib.LIB_PATH])
a = ap.parse_args()
flops = 2.0 * a.B * a.N * (a.C * a.C * a.E * 2 + (a.C * a.E) ** 2 * 3)
for path in a.libs:
    _lib._LIB = _lib.bind(C.CDLL(os.path.abspath(path)))
    torch.manual_seed(0)
    m = MixFFN(a.C, a.E).cuda()
    x = torch.randn(a.B, a.N, a.C, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            y = m(x)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters):
            y = m(x)
        e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / a.iters
    print(f"{os.path.basename(path):32s} {t * 1e3:8.1f} us  {flops / t / 1e9:7.1f} TFLOP/s  checksum {float(y.double().abs().sum()):.6e}")
