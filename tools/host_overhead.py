#!/usr/bin/env python3
"""Development: host time to ENQUEUE one forward step vs GPU time per step (is the step host-bound?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from efficientspeech_amd import CONFIGS, build_phoneme2mel, load_numpy_state_dict
from efficientspeech_amd.synth import synth_state_dict, synth_phonemes
from efficientspeech_amd.sharded import ShardedMelPipeline
cfg = CONFIGS["tiny"]; B, T = int(os.environ.get("HB", "256")), 128
net = build_phoneme2mel(cfg); load_numpy_state_dict(net, synth_state_dict(cfg, 1234)); net = net.cuda()
ids, mask = synth_phonemes(B, T, 1234)
x = {"phoneme": torch.from_numpy(ids).cuda(), "phoneme_mask": torch.from_numpy(mask).cuda(),
     "duration_forced": torch.full((B, T), 6, dtype=torch.int32, device="cuda"), "max_mel_len": 768}
pipe = ShardedMelPipeline(net, world_size=1, gather=False)
with torch.no_grad():
    for _ in range(20): pipe.step(x)
    torch.cuda.synchronize()
    N = 200
    t0 = time.perf_counter()
    for _ in range(N): pipe.step(x)
    t1 = time.perf_counter()
    pipe.flush(); torch.cuda.synchronize()
    t2 = time.perf_counter()
print("host enqueue per step: %.1f us   total per step: %.1f us" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
with torch.no_grad():
    for _ in range(100): pipe.step(x)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
