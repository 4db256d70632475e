#!/usr/bin/env python3
"""Development: shader-clock phase stamps of convgemm_dma_kernel (needs a -DESMI_GEMM_TRACE build: tools/build_variants.sh gtrace
"-DESMI_GEMM_TRACE").  Runs the dense k3 convolution of a MixFFN at base-ES block-1 size and prints, per wave of one mid-grid
workgroup, the mean cycles of each phase of a K chunk.   python tools/trace_gemm.py tools/_abl/libesmi_gtrace.so"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from efficientspeech_amd import _lib
from efficientspeech_amd.networks import MixFFN
lib = C.CDLL(os.path.abspath(sys.argv[1])); _lib._LIB = _lib.bind(lib)
lib.esmi_dev_set_gemm_trace.argtypes = [C.c_void_p]
from efficientspeech_amd import CONFIGS, build_phoneme2mel, load_numpy_state_dict
from efficientspeech_amd.synth import synth_state_dict, synth_phonemes
if len(sys.argv) > 2 and sys.argv[2] == "base":   # whole base-ES forward: the block-1 MixFFN convolution with its packed weight blob
    cfg = CONFIGS["base"]; B, T = 512, 256
    net = build_phoneme2mel(cfg); load_numpy_state_dict(net, synth_state_dict(cfg)); net = net.cuda()
    ids, mask = synth_phonemes(B, T, 1)
    x = {"phoneme": torch.from_numpy(ids).cuda(), "phoneme_mask": torch.from_numpy(mask).cuda(),
         "duration_forced": torch.full((B, T), 6, dtype=torch.int32, device="cuda"), "max_mel_len": 6 * T}
    m = net
else:
    Cc, E, N, B = 256, 2, 128, 512
    torch.manual_seed(0)
    m = MixFFN(Cc, E).cuda(); x = torch.randn(B, N, Cc, device="cuda")
NAMES = ["A frags read+split", "loads issued", "products issued", "wait loads", "stage W", "barrier", "loop back"]
with torch.no_grad():
    for _ in range(3): m(x)
    tr = torch.zeros((4, 512), dtype=torch.int64, device="cuda")
    lib.esmi_dev_set_gemm_trace(tr.data_ptr()); m(x); torch.cuda.synchronize(); lib.esmi_dev_set_gemm_trace(None)
t = tr.cpu().numpy()          # (only the k = 3 convolution records)
for w in range(4):
    v = t[w]; n = int((v != 0).sum()); v = v[:n - n % 7].reshape(-1, 7)
    d = np.diff(np.concatenate([v.reshape(-1), v[-1:, -1]]))[: v.size].reshape(-1, 7)
    print(f"wave {w}: {v.shape[0]} chunks, {(v[-1, -1] - v[0, 0]) / v.shape[0]:.0f} cycles per chunk")
    for j, nm in enumerate(NAMES):
        per_tap = "  ".join(f"{d[1 + r:-1:3, j].mean():6.0f}" for r in range(3))   # chunk index mod 3 = 1, 2, 0
        print(f"    {nm:22s} mean {d[1:-1, j].mean():8.0f}   min {d[1:-1, j].min():6d}  max {d[1:-1, j].max():6d}   by (it mod 3 = 1, 2, 0): {per_tap}")
