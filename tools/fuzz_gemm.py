#!/usr/bin/env python3
"""Development: MixFFN.forward (Linear -> dense k = 3 conv + GELU -> Linear: three launches of the LDS-staged GEMM of the per-op plan)
against a torch fp64 restatement on shapes chosen to hit the kernel's edges -- one-position utterances, utterances shorter than the
taps' reach, row counts that are not a multiple of the 128 / 256-row workgroup tile, channel counts that are not a multiple of 128.
python tools/fuzz_gemm.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from efficientspeech_amd.networks import MixFFN

torch.manual_seed(0)
worst = 0.0
cases = [(2048, 1, 128, 1), (1100, 2, 128, 2), (700, 3, 160, 1), (300, 7, 128, 2), (63, 33, 256, 1), (33, 64, 96, 2), (21, 100, 128, 1),
         (9, 255, 160, 2), (5, 513, 128, 1), (4100, 1, 96, 1), (17, 129, 224, 1), (2, 1500, 128, 2)]
for B, N, C, E in cases:
    m = MixFFN(C, E).cuda()
    x = torch.randn(B, N, C, device="cuda")
    with torch.no_grad():
        y = m(x)
        xd = x.double()
        h = F.linear(xd, m.mlp1.weight.double(), m.mlp1.bias.double())
        h = F.gelu(F.conv1d(h.transpose(1, 2), m.conv.weight.double(), m.conv.bias.double(), padding=1).transpose(1, 2))
        ref = F.linear(h, m.mlp2.weight.double(), m.mlp2.bias.double())
    err = float((y.double() - ref).abs().max()) / max(1e-9, float(ref.abs().max()))
    worst = max(worst, err)
    print(f"B={B:5d} N={N:5d} C={C:4d} E={E}: rows {B * N:6d}  max rel err {err:.2e}  finite {bool(torch.isfinite(y).all())}")
    assert err < 2e-5 and torch.isfinite(y).all(), "MISMATCH"
print(f"all {len(cases)} shapes within 2e-5 (worst {worst:.2e})")
