#!/usr/bin/env python3
"""Development: time the training step (efficientspeech_amd.train.TrainStep) on a synthetic teacher-forced batch.
python tools/bench_train.py [--config tiny] [--batch 32] [--phonemes 128] [--dur 6] [--iters 5]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from efficientspeech_amd import CONFIGS, build_phoneme2mel, train
from efficientspeech_amd.synth import synth_state_dict


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="tiny"); ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--phonemes", type=int, default=128); ap.add_argument("--dur", type=int, default=6); ap.add_argument("--iters", type=int, default=5); ap.add_argument("--graph", action="store_true"); ap.add_argument("--precision", type=int, default=32); ap.add_argument("--torch-mirror", action="store_true", help="time plain PyTorch-ROCm ops (tests/torch_mirror.py + torch.optim.AdamW) on the same batch instead")
    a = ap.parse_args()
    cfg = CONFIGS[a.config]
    net = build_phoneme2mel(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 1234).items()})
    net = net.cuda().train()
    x, y = train.synthetic_batch(a.batch, a.phonemes, a.dur, "cuda")
    if a.torch_mirror:
        from tests import torch_mirror as M
        for k, p in net.named_parameters():
            p.requires_grad_(not k.endswith("_bins"))
        opt = torch.optim.AdamW([p for p in net.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-6)

        class _S:
            def step(self, x, y):
                opt.zero_grad(set_to_none=True)
                parts, total = M.loss(M.train_forward(net, dict(x, mel=y["mel"])), x, y)
                total.backward()
                opt.step()
                return torch.stack([*[p.detach() for p in parts], total.detach()])
        step = _S()
    else:
        step = train.TrainStep(net, graph=a.graph, precision=a.precision, init_scale=2048.0)
    for _ in range(3):
        l0 = step.step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.iters):
        l1 = step.step(x, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
    frames = a.batch * a.phonemes * a.dur
    mode = " (plain PyTorch-ROCm ops)" if a.torch_mirror else (" (hipGraph)" if a.graph else "")
    print(f"train step{mode} {a.config}: B={a.batch} T={a.phonemes} L={a.phonemes * a.dur}: {dt * 1e3:.1f} ms/step  {frames / dt:.3e} mel-frames/s  "
          f"loss {float(l0[4]):.3f} -> {float(l1[4]):.3f}")
