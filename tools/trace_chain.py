#!/usr/bin/env python3
"""Development tool: shader-clock phase stamps of the chain kernels (needs a -DESMI_CHAIN_TRACE build)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from efficientspeech_amd import CONFIGS, _lib, build_phoneme2mel, load_numpy_state_dict
from efficientspeech_amd.synth import synth_state_dict, synth_phonemes
lib = C.CDLL(os.path.abspath(sys.argv[1])); _lib._LIB = _lib.bind(lib)
cfg = CONFIGS["tiny"]; B, T = 256, 128
net = build_phoneme2mel(cfg); load_numpy_state_dict(net, synth_state_dict(cfg)); net = net.cuda()
ids, mask = synth_phonemes(B, T, 1)
x = {"phoneme": torch.from_numpy(ids).cuda(), "phoneme_mask": torch.from_numpy(mask).cuda(),
     "duration_forced": torch.full((B, T), 6, dtype=torch.int32, device="cuda"), "max_mel_len": 768}
tr = torch.zeros((6, 64), dtype=torch.int64, device="cuda")
setters = [getattr(lib, "esmi_dev_set_chain_trace_" + tu) for tu in ("enc_attn_ffn", "enc_block", "enc_fuse_va", "enc_merge", "enc_va16", "enc_block16") if hasattr(lib, "esmi_dev_set_chain_trace_" + tu)]
for f in setters: f.argtypes = [C.c_void_p]
plan = int(sys.argv[2]) if len(sys.argv) > 2 else _lib.FUSE_ALL
ctx = _lib.launch_plan(plan); ctx.__enter__()
for _ in range(3): net(x)
for f in setters: f(tr.data_ptr())
net(x); torch.cuda.synchronize()
t = tr.cpu().numpy()
for slot, name in ((0, "E2 blk0 (NC=1)"), (1, "E2 blk1 (NC=2)"), (2, "E3 fuse+VA"), (3, "E1 blk0 merge+qkv"), (4, "E1 blk1 merge+qkv")):
    v = t[slot]; n = int((v != 0).sum())
    print(name, "total", v[n - 1] - v[0], "deltas", np.diff(v[:n]).tolist())
