#!/usr/bin/env python3
"""Generate tests/golden/hifigan_*.npz from the REFERENCE HiFi-GAN generator (runs only in the build container, where
/root/reference exists; the GPU box sees the .npz files only).

Weights = efficientspeech_amd.hifigan.synth_hifigan_state_dict (seeded PCG64 by key name), loaded into the reference's
`hifigan.Generator` AFTER remove_weight_norm() (the state a Lightning checkpoint holds, model.py:44).  Each fixture stores the
channels-last mel input and the waveform the reference produces, plus the config fields; a second output is produced from the
weight-norm parametrisation (weight_g / weight_v with a random positive g) to pin `fold_weight_norm`.
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
import hifigan as ref_hifigan  # noqa: E402  (the reference package)

from efficientspeech_amd.hifigan import HIFIGAN_CONFIGS, synth_hifigan_state_dict, hifigan_state_dict_spec  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def ref_generator(h):
    cfg = ref_hifigan.AttrDict(dict(resblock=h.resblock, upsample_rates=list(h.upsample_rates),
                                    upsample_kernel_sizes=list(h.upsample_kernel_sizes),
                                    upsample_initial_channel=h.upsample_initial_channel,
                                    resblock_kernel_sizes=list(h.resblock_kernel_sizes),
                                    resblock_dilation_sizes=[list(d) for d in h.resblock_dilation_sizes]))
    return ref_hifigan.Generator(cfg).eval()


def main():
    torch.manual_seed(0)
    for name, B, L in (("v2", 2, 24), ("v3", 1, 17), ("v1", 1, 9)):
        h = HIFIGAN_CONFIGS[name]
        sd = synth_hifigan_state_dict(h, 1234)
        g = ref_generator(h)
        wn_keys = list(g.state_dict().keys())                      # weight-norm form: *.weight_g / *.weight_v
        g.remove_weight_norm()
        plain = g.state_dict()
        # (registration order differs -- remove_weight_norm re-registers `weight` after `bias` -- names and shapes must agree)
        assert sorted((k, tuple(v.shape)) for k, v in plain.items()) == sorted(hifigan_state_dict_spec(h)), "key table mismatch"
        g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        rng = np.random.default_rng(7)
        mel = rng.standard_normal((B, L, h.num_mels)).astype(np.float32) * 2.0 - 4.0     # log-mel-like range
        with torch.no_grad():
            wav = g(torch.from_numpy(mel).transpose(1, 2)).squeeze(1).numpy()
        # the same weights expressed in weight-norm form with a random g: v = w * s, g = ||w|| (so that g v / ||v|| = w)
        wn = {}
        for k, v in sd.items():
            if k.endswith(".weight"):
                base = k[:-len(".weight")]
                w = torch.from_numpy(v)
                s = torch.from_numpy(rng.uniform(0.5, 2.0, size=(w.shape[0], 1, 1)).astype(np.float32))
                wn[base + ".weight_v"] = (w * s).numpy()
                wn[base + ".weight_g"] = w.reshape(w.shape[0], -1).norm(dim=1).reshape(-1, 1, 1).numpy()
            else:
                wn[k] = v
        assert sorted(wn) == sorted(wn_keys)
        np.savez_compressed(os.path.join(OUT, f"hifigan_{name}_b{B}_l{L}.npz"), config=name, mel=mel, wav=wav,
                            **{"wn." + k: v for k, v in wn.items() if name == "v2" and (k.startswith("conv_post") or k.startswith("ups.3"))})
        print(name, mel.shape, wav.shape, float(np.abs(wav).max()), float(np.abs(wav).mean()))


if __name__ == "__main__":
    main()
