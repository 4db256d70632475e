import sys, numpy as np, torch
sys.path.insert(0, ".")
from efficientspeech_amd import CONFIGS, build_phoneme2mel, load_numpy_state_dict
from efficientspeech_amd.synth import synth_state_dict, synth_phonemes
cfg = CONFIGS["tiny"]; B, T = 256, 128
net = build_phoneme2mel(cfg); load_numpy_state_dict(net, synth_state_dict(cfg)); net = net.cuda().eval()
ids, mask = synth_phonemes(B, T, 1)
rng = np.random.default_rng(1234); d = rng.integers(1, 12, size=(B, T)).astype(np.int32); L = int(d.sum(1).max())
x = {"phoneme": torch.from_numpy(ids).cuda(), "phoneme_mask": torch.from_numpy(mask).cuda(), "duration_forced": torch.from_numpy(d).cuda(), "max_mel_len": L}
with torch.no_grad():
    for _ in range(8): net(x)
torch.cuda.synchronize()
