#!/usr/bin/env python3
"""Development: 300 training steps over four batch shapes, eager and as hipGraphs (one per shape): loss curve, finiteness, wall time."""
import sys, time; sys.path.insert(0, "/root/repo")
import torch
from efficientspeech_amd import CONFIGS, build_phoneme2mel, train
from efficientspeech_amd.synth import synth_state_dict
cfg = CONFIGS["tiny"]; net = build_phoneme2mel(cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 1234).items()}); net = net.cuda().train()
batches = [train.synthetic_batch(16, 60 + 7 * i, 4, "cuda", seed=i) for i in range(4)]
for graph in (False, True):
    step = train.TrainStep(net, lr=1e-3, graph=graph)
    t0 = time.time(); hist = []
    for it in range(300):
        x, y = batches[it % 4]
        l = step.step(x, y)
        if it % 50 == 0 or it == 299: hist.append(round(float(l[4]), 3))
    torch.cuda.synchronize()
    print("graph" if graph else "eager", "300 steps over 4 batch shapes:", hist, "finite", bool(torch.isfinite(step.flat.data).all()), "%.1f s" % (time.time() - t0))
