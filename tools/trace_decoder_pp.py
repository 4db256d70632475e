#!/usr/bin/env python3
"""Development tool: per-quarter shader-clock timeline of one persistent workgroup of mel_decoder_pp_kernel
(needs a -DESMI_DEC_TRACE build).   python tools/trace_decoder_pp.py <lib.so>"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from efficientspeech_amd import CONFIGS, _lib, build_phoneme2mel, load_numpy_state_dict
from efficientspeech_amd.synth import synth_state_dict
lib = C.CDLL(os.path.abspath(sys.argv[1]))
_lib._LIB = _lib.bind(lib)
cfg = CONFIGS["tiny"]; B, T, D = 256, 128, 6; L = T * D
net = build_phoneme2mel(cfg); load_numpy_state_dict(net, synth_state_dict(cfg)); net = net.cuda()
feat = torch.randn((B, T, cfg.d4), device="cuda")
cum = (torch.arange(1, T + 1, device="cuda", dtype=torch.int32) * D).repeat(B, 1).contiguous()
mel_len = torch.full((B,), L, dtype=torch.int32, device="cuda")
h0 = torch.randn((B, T, cfg.dx2), device="cuda")
tr = torch.zeros((8, 512), dtype=torch.int64, device="cuda")
lib.esmi_dev_set_trace.argtypes = [C.c_void_p]
for _ in range(3):
    net.decoder._fused(feat, cum, mel_len, None, L, True, L, h0=h0)
lib.esmi_dev_set_trace(tr.data_ptr())
net.decoder._fused(feat, cum, mel_len, None, L, True, L, h0=h0)
torch.cuda.synchronize()
t = tr.cpu().numpy()
t0 = t[0, 0]
print("stamps: pairs (arrive, leave) per barrier; 4 barriers per step; work = leave[k-1] -> arrive[k], wait = arrive[k] -> leave[k]")
for w in (0, 4):
    n = int((t[w] > 0).sum()) // 2
    arr, lea = t[w, 0:2 * n:2], t[w, 1:2 * n:2]
    print(f"wave {w}: {n} barriers, total {lea[-1] - arr[0]} cycles")
    for st in range(n // 4):
        seg = []
        for q in range(4):
            k = 4 * st + q
            work = arr[k] - (lea[k - 1] if k > 0 else arr[0])
            seg.append(f"{int(work):6d}+{int(lea[k] - arr[k]):5d}")
        print(f"  step {st:2d} [start {int(arr[4 * st] - t0):8d}]: " + "  ".join(seg))
