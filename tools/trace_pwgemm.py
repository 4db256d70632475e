#!/usr/bin/env python3
"""Development: shader-clock phase stamps of pwgemm_kernel (needs a -DESMI_GEMM_TRACE build: tools/build_variants.sh gtrace
"-DESMI_GEMM_TRACE").  Runs the training step's decoder Linear (B x L rows, 128 -> 128, bias + ReLU) through esmi_train_conv_fwd_f32
and prints, per wave of one mid-grid workgroup, the cycles of each phase.   python tools/trace_pwgemm.py tools/_abl/libesmi_gtrace.so [rows]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from efficientspeech_amd import _lib
lib = C.CDLL(os.path.abspath(sys.argv[1])); _lib._LIB = _lib.bind(lib)
lib.esmi_dev_set_gemm_trace.argtypes = [C.c_void_p]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 76800
cin = cout = 128
torch.manual_seed(0)
x = torch.randn(1, rows, cin, device="cuda"); w = torch.randn(cout, cin, device="cuda") * 0.1; b = torch.randn(cout, device="cuda")
y = torch.empty(1, rows, cout, device="cuda")
d = _lib.ConvDesc(1, rows, cin, rows, cout, 1, 1, 0, 1, 0, 0, 1)
nws = lib.esmi_train_conv_workspace_bytes(C.byref(d)); ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
run = lambda: lib.esmi_train_conv_fwd_f32(C.byref(d), p(x), p(w), p(b), p(y), p(ws), nws, None)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("rows %d: %.1f us per launch; max err vs torch %.2e" % (rows, e0.elapsed_time(e1) * 50, (torch.relu(x[0] @ w.T + b) - y[0]).abs().max().item()))
tr = torch.zeros((8, 64), dtype=torch.int64, device="cuda")
lib.esmi_dev_set_gemm_trace(tr.data_ptr()); run(); torch.cuda.synchronize(); lib.esmi_dev_set_gemm_trace(None)
t = tr.cpu().numpy()
NAMES = ["rows arrived + split", "next loads issued", "products issued", "epilogue issued"]
HEAD = ["first rows issued", "weight loads issued", "everything arrived", "weight written", "barrier"]
for w_ in range(8):
    v = t[w_]; n = int((v != 0).sum())
    if n < 7: continue
    print(f"wave {w_}: " + "  ".join(f"{nm} {int(v[1 + k] - v[k])}" for k, nm in enumerate(HEAD)) + f"; {(n - 6) // 4} items; total {v[n - 1] - v[0]} cycles")
    for k in range((n - 6) // 4):
        s_ = v[6 + 4 * k: 10 + 4 * k]; prev = v[5 + 4 * k]
        print("    item %d: " % k + "  ".join(f"{nm} {int(a - b_)}" for nm, a, b_ in zip(NAMES, s_, [prev, *s_[:-1]])))
