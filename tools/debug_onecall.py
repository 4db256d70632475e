#!/usr/bin/env python3
"""Development: one-call forward vs per-stage path on a golden fixture; host enqueue cost; oracle thread scaling."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests import helpers as H
from efficientspeech_amd import _lib
g = np.load(os.path.join(ROOT, "tests/golden/tiny_eval_b1_fox.npz"))
net, cfg, sd = H.make_net("tiny", "cuda", golden=g)
x = H.to_x(g, "cuda")
with torch.no_grad():
    for rep in range(3):
        enc = net.encoder._encode(x)
        st = net._launch(x)
        torch.cuda.synchronize()
        print("rep", rep, "dur pred max diff one-call vs golden", np.abs(st.duration.cpu().numpy() - g["duration"]).max(),
              "per-stage vs golden", np.abs(enc["duration"].cpu().numpy() - g["duration"]).max(),
              "mel diff", np.abs(st.mel.cpu().numpy() - g["mel"]).max() if st.mel.shape == g["mel"].shape else st.mel.shape)
# host enqueue cost
from efficientspeech_amd.synth import synth_phonemes
net2, cfg, sd = H.make_net("tiny", "cuda")
for B in (32, 256):
    ids, mask = synth_phonemes(B, 128, 1)
    xx = {"phoneme": torch.from_numpy(ids).cuda(), "phoneme_mask": torch.from_numpy(mask).cuda(),
          "duration_forced": torch.full((B, 128), 6, dtype=torch.int32, device="cuda"), "max_mel_len": 768, "max_mel_len_exact": True}
    with torch.no_grad():
        for _ in range(20): net2(xx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): net2(xx)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"B={B}: enqueue {1e6*(t1-t0)/200:.1f} us/step, with drain {1e6*(t2-t0)/200:.1f} us/step")
# oracle thread scaling
from oracle import oracle
import ctypes
w = oracle.Weights(sd)
gomp = ctypes.CDLL("libgomp.so.1")
for n in (24, 64, 128, 256):
    gomp.omp_set_num_threads(n)
    ids, mask = synth_phonemes(64, 128, 99)
    d = np.full(ids.shape, 6, np.int32); z = np.zeros(ids.shape, np.float32)
    oracle.phoneme2mel(cfg, w, ids[:4], mask[:4], pitch=z[:4], energy=z[:4], duration=d[:4], f32=True)
    t0 = time.perf_counter()
    o = oracle.phoneme2mel(cfg, w, ids, mask, pitch=z, energy=z, duration=d, f32=True)
    dt = time.perf_counter() - t0
    print(f"oracle threads {n}: B=64 in {dt:.3f} s = {o.mel_len.sum()/dt:.3e} frames/s")
