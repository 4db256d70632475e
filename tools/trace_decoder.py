#!/usr/bin/env python3
"""Development tool: per-phase shader-clock breakdown of one mel-decoder workgroup (needs a -DESMI_DEC_TRACE build)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from efficientspeech_amd import CONFIGS, _lib, build_phoneme2mel, load_numpy_state_dict
from efficientspeech_amd.synth import synth_state_dict
lib = C.CDLL(os.path.abspath(sys.argv[1]))
_lib._LIB = _lib.bind(lib)
cname = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "tiny"
cfg = CONFIGS[cname]; B, T, D = (256, 128, 6) if cname == "tiny" else (256 if cname == "small" else 512, 256, 6); L = T * D
net = build_phoneme2mel(cfg); load_numpy_state_dict(net, synth_state_dict(cfg)); net = net.cuda()
feat = torch.randn((B, T, cfg.d4), device="cuda")
cum = (torch.arange(1, T + 1, device="cuda", dtype=torch.int32) * D).repeat(B, 1).contiguous()
mel_len = torch.full((B,), L, dtype=torch.int32, device="cuda")
h0 = torch.randn((B, T, cfg.dx2), device="cuda") if "--h0" in sys.argv else None
tr = torch.zeros((8, 64), dtype=torch.int64, device="cuda")
lib.esmi_dev_set_trace.argtypes = [C.c_void_p]
for _ in range(3):
    net.decoder._fused(feat, cum, mel_len, None, L, True, L, h0=h0)
lib.esmi_dev_set_trace(tr.data_ptr())
net.decoder._fused(feat, cum, mel_len, None, L, True, L, h0=h0)
torch.cuda.synchronize()
t = tr.cpu().numpy()
names = ["dw window load", "barrier", "dw compute+write", "barrier", "K loop (MFMA)", "barrier", "bias+tanh store", "barrier", "LayerNorm", "barrier"]
nl = min(cfg.n_blocks * cfg.block_depth, 5)      # (64 stamp slots per wave = 5 layers of 11 stamps)
for w in (0, 3, 7):
    print(f"wave {w}: total layer-loop cycles {t[w, 11 * nl - 1] - t[w, 0]}")
    for l in range(nl):
        d = np.diff(t[w, 11 * l: 11 * l + 11])
        print(f"  layer {l}: " + "  ".join(f"{n}={int(x)}" for n, x in zip(names, d)))
