#!/usr/bin/env python3
"""Golden vectors of the TRAINING step (model.py:167-226), from the reference in the build container.

The reference's Phoneme2Mel (train=True) on the seeded synthetic weights and a seeded teacher-forced batch, the reference's
loss arithmetic (model.py:167-209: masked L1 on mel, masked MSE on pitch / energy / log(duration + 1), weights 10 / 2 / 2 / 1,
model.py:216), torch autograd for the gradients and torch.optim.AdamW (model.py:279-283 with its lr / weight_decay defaults)
for one update.  The fixture holds data only: inputs, the four losses, the total, every parameter's gradient, and the
parameters after one optimizer step.   python tools/gen_golden_train.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G   # noqa: E402  (reference import + stubs + build_ref)

from efficientspeech_amd.config import CONFIGS          # noqa: E402
from efficientspeech_amd.synth import synth_phonemes, _rng   # noqa: E402


def reference_loss(y_hat, mel, x):
    """The arithmetic of model.py:167-209 on already-selected tensors."""
    mm = (~x["mel_mask"]).unsqueeze(-1)
    mel_loss = nn.L1Loss()(y_hat["mel"].masked_select(mm), mel.masked_select(mm))
    pm = ~x["phoneme_mask"]
    out = [mel_loss]
    for k in ("pitch", "energy"):
        out.append(nn.MSELoss()(torch.squeeze(y_hat[k]).masked_select(pm), x[k].masked_select(pm)))
    d, dp = x["duration"].masked_select(pm), torch.squeeze(y_hat["duration"]).masked_select(pm)
    out.append(nn.MSELoss()(torch.log(dp.float() + 1), torch.log(d.float() + 1)))
    return out


def main():
    outdir = os.path.join(G.ROOT, "tests", "golden")
    for name, lens, T in (("tiny", [13, 9], 13), ("tiny", [11], 11), ("small", [13, 9], 13), ("base", [13, 9], 13)):
        cfg = CONFIGS[name]
        net, sd = G.build_ref(cfg)
        net.train()
        B = len(lens)
        ph, m = synth_phonemes(B, T, G.SEED + 7, lens)
        g = _rng(G.SEED, "train-targets")
        pitch = g.uniform(-3.5, 12.0, size=(B, T)).astype(np.float32)
        energy = g.uniform(-2.0, 9.0, size=(B, T)).astype(np.float32)
        dur = g.integers(1, 7, size=(B, T)).astype(np.int32)
        if B == 1:
            dur[0, [2, 7]] = 0                      # zero-length phonemes inside the utterance
        dur[m] = 0
        mel_len = dur.sum(1).astype(np.int32)
        L = int(mel_len.max())
        mel = g.normal(-5.0, 2.0, size=(B, L, 80)).astype(np.float32)
        mel_mask = np.arange(L)[None, :] >= mel_len[:, None]
        x = dict(phoneme=torch.from_numpy(ph), phoneme_mask=torch.from_numpy(m), pitch=torch.from_numpy(pitch),
                 energy=torch.from_numpy(energy), duration=torch.from_numpy(dur).long(), mel_len=torch.from_numpy(mel_len),
                 mel_mask=torch.from_numpy(mel_mask))
        params = [p for p in net.parameters() if p.requires_grad]
        opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-6)      # model.py:107-108 defaults, :280
        opt.zero_grad()
        y_hat = net(x, train=True)
        losses = reference_loss(y_hat, torch.from_numpy(mel), x)
        total = 10.0 * losses[0] + 2.0 * losses[1] + 2.0 * losses[2] + losses[3]      # model.py:216
        total.backward()
        rec = dict(in_phoneme=ph, in_phoneme_mask=m, in_pitch=pitch, in_energy=energy, in_duration=dur, in_mel_len=mel_len,
                   in_mel=mel, in_mel_mask=mel_mask, losses=np.array([float(v) for v in losses], np.float64),
                   total=np.array(float(total), np.float64), weights_crc=np.array(G.sd_crc(sd), dtype=np.uint32),
                   mel_pred=y_hat["mel"].detach().numpy() if name == "tiny" else y_hat["mel"].detach().numpy()[:, :8], pitch_pred=y_hat["pitch"].detach().numpy(),
                   energy_pred=y_hat["energy"].detach().numpy(), duration_pred=y_hat["duration"].detach().numpy())
        names = [k for k, p in net.named_parameters() if p.requires_grad]
        nograd = [k for k, p in net.named_parameters() if p.requires_grad and p.grad is None]
        sample = ("encoder.encoder.embed.weight", "encoder.encoder.attn_blocks.0.2.qkv.weight", "encoder.fuse.mlps.1.1.weight",
                  "encoder.duration_decoder.conv1.0.weight", "encoder.pitch_decoder.pitch_embedding.weight",
                  "decoder.blocks.0.0.0.0.0.weight", "decoder.blocks.1.1.bias", "decoder.mel_linear.weight")
        for k, p in net.named_parameters():
            full = B > 1 and name == "tiny"                                          # every gradient for the headline config; a sample
            small_ones = p.numel() <= 4096 and (k.endswith(".bias") or ".norm" in k or k.endswith("linear.weight"))   # elsewhere (file size)
            keep = k in sample if name == "tiny" else (small_ones or k in ("decoder.mel_linear.weight", "encoder.encoder.embed.weight"))
            if p.requires_grad and p.grad is not None and (full or keep):
                rec["grad." + k] = p.grad.numpy().astype(np.float32)
        opt.step()
        keep_after = ("encoder.encoder.embed.weight", "decoder.mel_linear.weight", "decoder.mel_linear.bias",
                      "encoder.duration_decoder.linear.weight", "encoder.fuse.conv.weight")      # AdamW is elementwise: a sample
        for k, p in net.named_parameters():
            if (k in keep_after and name == "tiny") or k.endswith("norm1.bias"):
                rec["after." + k] = p.detach().numpy().astype(np.float32)
        rec["param_names"] = np.array(names)
        rec["no_grad_params"] = np.array(nograd)
        path = os.path.join(outdir, f"{name}_train_step.npz" if B > 1 else f"{name}_train_step_b1.npz")
        if name != "tiny":
            rec.pop("in_mel_check", None)
        np.savez_compressed(path, **rec)
        print(path, os.path.getsize(path), "bytes; losses", rec["losses"], "total", float(total), "params without grad:", nograd)


def main_amp():
    """The same step as the reference trains it by default: `--precision 16` (utils/tools.py:326-327) = Lightning's
    torch.autocast(dtype=float16) around training_step + torch.amp.GradScaler around the optimizer (here on the CPU: the build
    container has no GPU).  -> tests/golden/tiny_train_step_amp16.npz: losses, every gradient after GradScaler.unscale_, a sample
    of the parameters after the scaled step, the scaler's scale."""
    outdir = os.path.join(G.ROOT, "tests", "golden")
    name, lens, T = "tiny", [13, 9], 13
    cfg = CONFIGS[name]
    net, sd = G.build_ref(cfg)
    net.train()
    B = len(lens)
    ph, m = synth_phonemes(B, T, G.SEED + 7, lens)
    g = _rng(G.SEED, "train-targets")
    pitch = g.uniform(-3.5, 12.0, size=(B, T)).astype(np.float32)
    energy = g.uniform(-2.0, 9.0, size=(B, T)).astype(np.float32)
    dur = g.integers(1, 7, size=(B, T)).astype(np.int32)
    dur[m] = 0
    mel_len = dur.sum(1).astype(np.int32)
    L = int(mel_len.max())
    mel = g.normal(-5.0, 2.0, size=(B, L, 80)).astype(np.float32)
    mel_mask = np.arange(L)[None, :] >= mel_len[:, None]
    x = dict(phoneme=torch.from_numpy(ph), phoneme_mask=torch.from_numpy(m), pitch=torch.from_numpy(pitch),
             energy=torch.from_numpy(energy), duration=torch.from_numpy(dur).long(), mel_len=torch.from_numpy(mel_len),
             mel_mask=torch.from_numpy(mel_mask))
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-6)
    # GradScaler's default init_scale (65536) overflows binary16 in this step's backward: the reference skips five steps on this batch
    # (65536 -> 2048) before its first update.  The fixture records that first clean step: init_scale = 2048.
    scaler = torch.amp.GradScaler("cpu", init_scale=2048.0)
    opt.zero_grad()
    with torch.autocast("cpu", dtype=torch.float16):
        y_hat = net(x, train=True)
        losses = reference_loss(y_hat, torch.from_numpy(mel), x)
        total = 10.0 * losses[0] + 2.0 * losses[1] + 2.0 * losses[2] + losses[3]
    scaler.scale(total).backward()
    scaler.unscale_(opt)
    rec = dict(in_phoneme=ph, in_phoneme_mask=m, in_pitch=pitch, in_energy=energy, in_duration=dur, in_mel_len=mel_len, in_mel=mel,
               in_mel_mask=mel_mask, losses=np.array([float(v) for v in losses], np.float64), total=np.array(float(total), np.float64),
               weights_crc=np.array(G.sd_crc(sd), dtype=np.uint32), out_dtype=np.array(str(y_hat["mel"].dtype)))
    big = ("encoder.encoder.embed.weight", "encoder.encoder.attn_blocks.0.2.qkv.weight", "encoder.fuse.mlps.1.1.weight",
           "encoder.duration_decoder.conv1.0.weight", "decoder.blocks.0.0.0.0.1.weight", "decoder.mel_linear.weight")
    for k, p in net.named_parameters():
        if p.requires_grad and p.grad is not None and (p.numel() <= 4096 or k in big):      # (file size: every small tensor, six large ones)
            rec["grad." + k] = p.grad.float().numpy()
    scaler.step(opt)
    scaler.update()
    for k, p in net.named_parameters():
        if k in ("encoder.encoder.embed.weight", "decoder.mel_linear.weight", "decoder.mel_linear.bias", "encoder.duration_decoder.linear.weight"):
            rec["after." + k] = p.detach().numpy().astype(np.float32)
    rec["scale_after"] = np.array(float(scaler.get_scale()))
    path = os.path.join(outdir, "tiny_train_step_amp16.npz")
    np.savez_compressed(path, **rec)
    print(path, os.path.getsize(path), "bytes; losses", rec["losses"], "total", float(total), "scale", float(scaler.get_scale()))


if __name__ == "__main__":
    if "--amp" in sys.argv:
        main_amp()
    else:
        main()
