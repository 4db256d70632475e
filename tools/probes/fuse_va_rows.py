#!/usr/bin/env python3
"""Development probe (profiles/r05_probes/fuse_va_wrong_rows.md): which rows of enc_fuse_va_kernel's outputs differ between the product
library and a variant build (tools/_abl/libesmi_<tag>.so, -DESMI_E3_STAGED=<rows per batch>)?  Launch plan 31 = the round-1..4 chain
kernels.  usage: fuse_va_rows.py <tag> [<tag> ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from efficientspeech_amd import CONFIGS, _lib, build_phoneme2mel, load_numpy_state_dict
from efficientspeech_amd.synth import synth_state_dict, synth_phonemes

cfg = CONFIGS["tiny"]
sd = synth_state_dict(cfg, 1234)


def run(lib_path, B, T, lens=None):
    ctx = _lib.use_library(lib_path) if lib_path else None
    if ctx:
        ctx.__enter__()
    try:
        net = build_phoneme2mel(cfg); load_numpy_state_dict(net, sd); net = net.cuda()
        ids, mask = synth_phonemes(B, T, 11, lens)
        x = {"phoneme": torch.from_numpy(ids).cuda()}
        if B > 1:
            x["phoneme_mask"] = torch.from_numpy(mask).cuda()
        with _lib.launch_plan(31), torch.no_grad():
            enc = net.encoder._encode(x, need_lmax=False)
        torch.cuda.synchronize()
        return {k: enc[k].float().cpu().numpy() for k in ("pitch", "energy", "duration", "feat")}
    finally:
        if ctx:
            ctx.__exit__(None, None, None)


for tag in sys.argv[1:]:
    path = os.path.join(ROOT, "tools", "_abl", f"libesmi_{tag}.so")
    for (B, T) in ((1, 32), (1, 31), (2, 64), (1, 128)):
        ref, var = run(None, B, T), run(path, B, T)
        out = []
        for k in ("pitch", "energy", "duration"):
            d = np.abs(ref[k] - var[k]).reshape(B, T)
            bad = sorted(set(np.argwhere(d > 1e-4)[:, 1].tolist()))
            out.append(f"{k}: rows {bad} max {d.max():.3g}")
        d = np.abs(ref["feat"] - var["feat"])[..., 96:].max(-1).reshape(B, T)          # duration features = LN2(conv2(hidden))
        bad = sorted(set(np.argwhere(d > 1e-4)[:, 1].tolist()))
        print(f"[{tag}] B={B} T={T}  " + " | ".join(out) + f" | dur feats: rows {bad} max {d.max():.3g}", flush=True)
