#!/usr/bin/env python3
"""Development: the HIP forward vs tests/torch_mirror.py (GPU and CPU) per utterance at three batch sizes -- finds bucket-edge flips."""
import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from efficientspeech_amd import CONFIGS, build_phoneme2mel
from efficientspeech_amd.synth import synth_state_dict, synth_phonemes
from tests import torch_mirror as M
cfg = CONFIGS["tiny"]; sd = synth_state_dict(cfg, 1234)
net = build_phoneme2mel(cfg); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net = net.cuda().eval()
for B, T in ((3, 20), (16, 128), (256, 128)):
    ids, mask = synth_phonemes(B, T, 1234)
    x = {"phoneme": torch.from_numpy(ids).cuda(), "phoneme_mask": torch.from_numpy(mask).cuda(),
         "duration_forced": torch.full((B, T), 6, dtype=torch.int32, device="cuda"), "max_mel_len": 6 * T, "max_mel_len_exact": True}
    with torch.no_grad():
        m = M.eval_forward(net, x)[0]; h = net(x)[0]
        mc = M.eval_forward(net.cpu(), {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in x.items()})[0]; net.cuda()
    d = (m - h).abs().amax(dim=(1, 2))
    print(B, T, "gpu mirror vs hip", float(d.max()), "worst utt", int(d.argmax()), "n_bad", int((d > 1e-4).sum()), "| cpu mirror vs hip", float((mc.cuda() - h).abs().max()))
