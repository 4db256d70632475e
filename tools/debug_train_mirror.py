#!/usr/bin/env python3
"""Development: every parameter gradient of the HIP training step against torch.autograd over tests/torch_mirror.py (float64) on a
ragged batch of a chosen size, for the matrix-pipe and the plain-fp32 operator paths.
python tools/debug_train_mirror.py [tiny|small|base] B T"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from efficientspeech_amd import CONFIGS, build_phoneme2mel, train
from efficientspeech_amd.synth import synth_state_dict, synth_phonemes
from tests import torch_mirror as M

name, B, T = (sys.argv + ["tiny", "6", "97"])[1], int((sys.argv + ["tiny", "6", "97"])[2]), int((sys.argv + ["tiny", "6", "97"])[3])
dev, cfg = "cuda", CONFIGS[name]
sd = synth_state_dict(cfg, 1234)


def mk():
    n = build_phoneme2mel(cfg)
    n.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return n.to(dev).train()


rng = np.random.default_rng(3)
lens = sorted(rng.integers(T // 3, T + 1, B).tolist(), reverse=True); lens[0] = T
ids, mask = synth_phonemes(B, T, 11, lens)
dur = rng.integers(0, 7, (B, T)).astype(np.int32); dur[mask] = 0
mel_len = dur.sum(1); L = int(mel_len.max())
t = lambda a: torch.from_numpy(a).to(dev)      # noqa: E731
x = {"phoneme": t(ids), "phoneme_mask": t(mask), "pitch": t(rng.uniform(-3, 11, (B, T)).astype(np.float32)),
     "energy": t(rng.uniform(-2, 8, (B, T)).astype(np.float32)), "duration": t(dur), "mel_len": t(mel_len.astype(np.int32)),
     "mel_mask": t(np.arange(L)[None, :] >= mel_len[:, None])}
y = {"mel": t(rng.normal(-5, 2, (B, L, 80)).astype(np.float32))}
ref = mk().double()
x64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in dict(x, mel=y["mel"]).items()}
_, t64 = M.loss(M.train_forward(ref, x64), x64, {"mel": y["mel"].double()})
t64.backward()
rg = {k: p.grad for k, p in ref.named_parameters()}
for pipe in (False, True):
    train.USE_MATRIX_PIPE = pipe
    n = mk()
    parts, total = train.training_loss(n, x, y)
    total.backward()
    errs = sorted(((float((p.grad.double() - rg[k]).abs().max()) / max(1e-12, float(rg[k].abs().max())), k)
                   for k, p in n.named_parameters() if rg[k] is not None), reverse=True)
    print(f"matrix pipe {pipe}: total {float(total):.6f} (fp64 mirror {float(t64):.6f}); tensors above 2e-5: {sum(e > 2e-5 for e, _ in errs)} of {len(errs)}; "
          f"median {errs[len(errs) // 2][0]:.1e}; worst " + ", ".join(f"{k} {e:.1e}" for e, k in errs[:3]))
