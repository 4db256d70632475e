#!/usr/bin/env python3
"""Development: time the HiFi-GAN generator (esmi_hifigan_generator_f32) alone.  python tools/bench_vocoder.py [--config v2] [--batch 16] [--frames 768]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from efficientspeech_amd.hifigan import HIFIGAN_CONFIGS, Generator, synth_hifigan_state_dict, flops_per_mel_frame

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="v2"); ap.add_argument("--batch", type=int, default=16); ap.add_argument("--frames", type=int, default=768)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
h = HIFIGAN_CONFIGS[a.config]


voc = Generator(h)
voc.load_state_dict({k: torch.from_numpy(v) for k, v in synth_hifigan_state_dict(h, 1234).items()})
voc = voc.cuda().eval()
mel = torch.randn((a.batch, a.frames, h.num_mels), device="cuda") * 2 - 4
with torch.no_grad():
    for _ in range(2):
        wav = voc(mel.transpose(1, 2))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        wav = voc(mel.transpose(1, 2))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
fl = flops_per_mel_frame(h)
frames = a.batch * a.frames
print(f"hifigan {a.config}: B={a.batch} L={a.frames}: {dt*1e3:.2f} ms  {frames/dt:.3e} mel-frames/s  {frames*h.hop/dt/22050:.1f}x real time  "
      f"{fl/1e6:.1f} MFLOP/frame -> {fl*frames/dt/1e12:.1f} TFLOP/s   finite={bool(torch.isfinite(wav).all())}")
