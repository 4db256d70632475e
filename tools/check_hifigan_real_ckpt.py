#!/usr/bin/env python3
"""Build-container check (needs /root/reference): the reference's own LJ_V2/generator_v2 checkpoint through fold_weight_norm and
the oracle's HiFi-GAN restatement vs the reference Generator.  Last run: max |diff| 8.1e-07 (wav absmax 0.19)."""
import json, sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.path.insert(0, "/root/reference")
import hifigan as rh
from efficientspeech_amd.hifigan import fold_weight_norm, HifiGanConfig
from oracle import oracle
cfg = json.load(open("/root/reference/hifigan/LJ_V2/config.json"))
g = rh.Generator(rh.AttrDict(cfg)); ck = torch.load("/root/reference/hifigan/LJ_V2/generator_v2", map_location="cpu")
g.load_state_dict(ck["generator"]); g.eval(); g.remove_weight_norm()
fold = fold_weight_norm(ck["generator"])
for k, v in g.state_dict().items():
    assert np.allclose(v.numpy(), fold[k].numpy(), rtol=1e-6, atol=1e-8), k
mel = (np.random.default_rng(0).standard_normal((1, 20, 80)) * 2 - 5).astype(np.float32)
with torch.no_grad():
    ref = g(torch.from_numpy(mel).transpose(1, 2)).squeeze(1).numpy()
out = oracle.hifigan(HifiGanConfig.from_json(cfg), oracle.Weights({k: v.numpy() for k, v in fold.items()}), mel)
print("real LJ_V2 weights: oracle vs reference max diff", np.abs(out - ref).max(), "wav absmax", np.abs(ref).max())
