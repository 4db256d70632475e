#!/usr/bin/env python3
"""Experiment: G independent sub-batch pipelines (B / G utterances each) on G streams against one B-utterance step on one stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from efficientspeech_amd import CONFIGS, build_phoneme2mel, load_numpy_state_dict
from efficientspeech_amd.synth import synth_state_dict, synth_phonemes
cfg = CONFIGS["tiny"]; B, T, D = 256, 128, 6
sd = synth_state_dict(cfg, 1234)
def mk():
    net = build_phoneme2mel(cfg); load_numpy_state_dict(net, sd); return net.cuda().eval()
def batch(b, seed):
    ids, mask = synth_phonemes(b, T, seed)
    return {"phoneme": torch.from_numpy(ids).cuda(), "phoneme_mask": torch.from_numpy(mask).cuda(),
            "duration_forced": torch.full((b, T), D, dtype=torch.int32, device="cuda"), "max_mel_len": T * D, "max_mel_len_exact": True}
steps = 200
for G in (1, 2, 4):
    nets = [mk() for _ in range(G)]                     # one module per stream: their packed caches / arenas are per-module
    xs = [batch(B // G, 10 + g) for g in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    with torch.no_grad():
        for it in range(steps + 20):
            if it == 20:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    nets[g](xs[g])
            if it % 4 == 3:                             # bound the host's run-ahead
                ev = [torch.cuda.Event() for _ in range(G)]
                for g in range(G): ev[g].record(streams[g])
                if it >= 8: prev[0].synchronize()
                prev = ev
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"G={G}: {dt * 1e3:.4f} ms per {B} utterances  {B * T * D / dt:.3e} frames/s")
