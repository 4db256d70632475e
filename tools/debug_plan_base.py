"""Development: ms/step of a config under different launch plans (ESMI_FUSE_* masks).  python tools/debug_plan_base.py [tiny|small|base]"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from efficientspeech_amd import CONFIGS, build_phoneme2mel, _lib
from efficientspeech_amd.synth import synth_state_dict
import numpy as np
name = sys.argv[1] if len(sys.argv) > 1 else "base"
cfg = CONFIGS[name]; sd = synth_state_dict(cfg, 1234)
net = build_phoneme2mel(cfg); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net = net.cuda().eval()
B, T = {"tiny": (256, 128), "small": (256, 256), "base": (512, 256)}[name]
ids = torch.from_numpy(np.random.default_rng(1234).integers(1, 153, size=(B, T)).astype(np.int32)).cuda()
x = {"phoneme": ids, "phoneme_mask": torch.zeros((B, T), dtype=torch.bool, device="cuda"), "duration_forced": torch.full((B, T), 6, dtype=torch.int32, device="cuda"), "max_mel_len": 6 * T, "max_mel_len_exact": True}
for mask in (31, 31 & ~2, 31 & ~1, 0):
    with _lib.launch_plan(mask), torch.no_grad():
        for _ in range(3): net(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): net(x)
        torch.cuda.synchronize(); print("plan", mask, "ms/step %.3f" % ((time.perf_counter() - t0) / 10 * 1e3))
