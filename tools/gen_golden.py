#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE implementation in the build container.

Runs only where /root/reference exists (never on the GPU box).  Imports the reference's
`layers` package (with the two-module stub of SURVEY.md §8c), loads the seeded synthetic
weights of efficientspeech_amd.synth with strict=True, runs the reference modules on seeded
inputs and writes small .npz fixtures to tests/golden/.  The fixtures hold data only
(inputs, expected outputs, stage taps); no reference source travels.

    python tools/gen_golden.py            # all configs
"""
import os
import sys
import types
import zlib

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

# --- stub the two absent text-front-end deps (never reached on the acoustic-model path)
_u = types.ModuleType("unidecode"); _u.unidecode = lambda s: s; sys.modules["unidecode"] = _u
class _E:
    def number_to_words(self, *a, **k): return ""
_i = types.ModuleType("inflect"); _i.engine = lambda: _E(); sys.modules["inflect"] = _i

from layers import PhonemeEncoder, MelDecoder, Phoneme2Mel          # noqa: E402  (reference)

from efficientspeech_amd.config import CONFIGS, LJSPEECH_PITCH_STATS, LJSPEECH_ENERGY_STATS  # noqa: E402
from efficientspeech_amd.synth import synth_state_dict, state_dict_spec, synth_phonemes, _rng  # noqa: E402

SEED = 1234
# 31 ARPAbet ids of "the quick brown fox ..." (probed from text_to_sequence)
FOX = [92, 74, 117, 145, 110, 117, 89, 131, 83, 120, 105, 67, 117, 132, 116, 75, 119, 130, 132, 124, 144, 98,
       92, 74, 118, 103, 147, 113, 91, 79, 106]


def build_ref(cfg):
    enc = PhonemeEncoder(pitch_stats=LJSPEECH_PITCH_STATS, energy_stats=LJSPEECH_ENERGY_STATS, **cfg.encoder_kwargs())
    dec = MelDecoder(**cfg.decoder_kwargs())
    net = Phoneme2Mel(enc, dec).eval()
    ref_spec = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert ref_spec == [(k, s) for k, s, _ in state_dict_spec(cfg)], "state_dict key table drifted"
    sd = synth_state_dict(cfg, SEED)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return net, sd


def sd_crc(sd):
    c = 0
    for k, v in sd.items():
        c = zlib.crc32(v.tobytes(), zlib.crc32(k.encode(), c))
    return c


def run_case(net, x, train):
    """Run reference Phoneme2Mel with hooks that tap the stage outputs."""
    taps = {}
    hs = []
    enc = net.encoder
    hs.append(enc.encoder.register_forward_hook(lambda m, i, o: taps.update(
        {f"f{j}": t.detach().numpy().copy() for j, t in enumerate(o[0])})))
    hs.append(enc.fuse.register_forward_hook(lambda m, i, o: taps.update(fused=o.detach().numpy().copy())))
    hs.append(enc.pitch_decoder.pitch_embedding.register_forward_pre_hook(
        lambda m, i: taps.update(pitch_idx=i[0].detach().numpy().reshape(taps["fused"].shape[:2]).astype(np.int32))))
    hs.append(enc.energy_decoder.energy_embedding.register_forward_pre_hook(
        lambda m, i: taps.update(energy_idx=i[0].detach().numpy().reshape(taps["fused"].shape[:2]).astype(np.int32))))
    hs.append(enc.feature_upsampler.register_forward_hook(lambda m, i, kw, o: taps.update(
        feat=i[0].detach().numpy().copy(),
        dur=np.stack([d.detach().numpy().reshape(-1) for d in kw["duration"]]).astype(np.int32),
        features=o[0].detach().numpy().copy(), masks=o[1].detach().numpy()[:, :, 0].copy()), with_kwargs=True))
    with torch.no_grad():
        out = net(x, train=train)
    for h in hs:
        h.remove()
    if train:
        mel, mel_len, dpred = out["mel"], out["mel_len"], out["duration"]
        taps["pitch"] = out["pitch"].numpy().copy()
        taps["energy"] = out["energy"].numpy().copy()
        taps["has_masks"] = np.array(out["masks"] is not None)
    else:
        mel, mel_len, dpred = out
    taps.update(mel=mel.numpy().copy(), mel_len=mel_len.numpy().astype(np.int32), duration=dpred.numpy().copy())
    return taps


def main():
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    for name, cfg in CONFIGS.items():
        net, sd = build_ref(cfg)
        common = dict(weights_crc=np.array(sd_crc(sd), dtype=np.uint32), seed=np.array(SEED),
                      pitch_bins=sd["encoder.pitch_decoder.pitch_bins"],
                      energy_bins=sd["encoder.energy_decoder.energy_bins"])
        cases = {}
        # A: config 1 -- single utterance, B==1 code path (no masks at all, networks.py:338)
        ph = np.array([FOX], dtype=np.int32)
        cases["eval_b1_fox"] = (dict(phoneme=ph), False)
        # B: padded batch, odd T (mask pooling pads with True, blocks.py:52-57)
        lens = [17, 11, 6]
        ph, m = synth_phonemes(3, 17, SEED, lens)
        cases["eval_pad_t17"] = (dict(phoneme=ph, phoneme_mask=m), False)
        # C: even T, no padding
        ph, m = synth_phonemes(4, 16, SEED)
        cases["eval_even_t16"] = (dict(phoneme=ph, phoneme_mask=m), False)
        # D: teacher-forced (train=True data flow): targets bucketised, durations forced
        ph, m = synth_phonemes(3, 17, SEED + 1, lens)
        g = _rng(SEED, "targets")
        pitch = g.uniform(-3.5, 12.0, size=(3, 17)).astype(np.float32)
        energy = g.uniform(-2.0, 9.0, size=(3, 17)).astype(np.float32)
        dur = g.integers(0, 8, size=(3, 17)).astype(np.int32)
        dur[m] = 0
        mel_len = dur.sum(1).astype(np.int32)
        cases["train_tf_t17"] = (dict(phoneme=ph, phoneme_mask=m, pitch=pitch, energy=energy, duration=dur,
                                      mel_len=mel_len), True)
        # D2: teacher-forced with max_mel_len larger than every utterance (pure padding tail)
        mel_len2 = mel_len.copy(); mel_len2[0] += 9
        cases["train_tf_padtail"] = (dict(phoneme=ph, phoneme_mask=m, pitch=pitch, energy=energy, duration=dur,
                                          mel_len=mel_len2), True)
        for cname, (x, train) in cases.items():
            xt = {k: torch.from_numpy(np.asarray(v)) for k, v in x.items()}
            if "duration" in xt:
                xt["duration"] = xt["duration"].long()
            taps = run_case(net, xt, train)
            if name != "tiny":                      # keep the fixtures small: features == gather(feat, dur)
                taps.pop("features")
            rec = dict(common)
            rec.update({"in_" + k: np.asarray(v) for k, v in x.items()})
            rec.update(taps)
            rec["train"] = np.array(train)
            path = os.path.join(outdir, f"{name}_{cname}.npz")
            np.savez_compressed(path, **rec)
            print(f"{path}: mel {taps['mel'].shape} mel_len {taps['mel_len'].tolist()} "
                  f"dur[0,:8] {taps['dur'][0, :8].tolist()} pidx[0,:8] {taps['pitch_idx'][0, :8].tolist()}")
        # E: forced-duration eval (D-const 6): predicted pitch/energy embeddings are kept by feeding the
        # eval predictions back as targets -- bucketize(pred) is then identical to the eval path.
        ph, m = synth_phonemes(2, 16, SEED + 2)
        xt = dict(phoneme=torch.from_numpy(ph), phoneme_mask=torch.from_numpy(m))
        ev = run_case(net, xt, False)
        dur = np.full((2, 16), 6, np.int32)
        x = dict(phoneme=ph, phoneme_mask=m, pitch=run_pred(net, xt, "pitch"), energy=run_pred(net, xt, "energy"),
                 duration=dur, mel_len=dur.sum(1).astype(np.int32))
        xt = {k: torch.from_numpy(np.asarray(v)) for k, v in x.items()}
        xt["duration"] = xt["duration"].long()
        taps = run_case(net, xt, True)
        assert np.array_equal(taps["pitch_idx"], ev["pitch_idx"]) and np.array_equal(taps["energy_idx"], ev["energy_idx"])
        if name != "tiny":
            taps.pop("features")
        rec = dict(common)
        rec.update({"in_" + k: np.asarray(v) for k, v in x.items()})
        rec.update(taps)
        rec["train"] = np.array(True)
        path = os.path.join(outdir, f"{name}_forced_d6_t16.npz")
        np.savez_compressed(path, **rec)
        print(f"{path}: mel {taps['mel'].shape}")


def run_pred(net, xt, which):
    with torch.no_grad():
        out = net.encoder(xt, train=False)
    return out[which].numpy()[..., 0].copy()


if __name__ == "__main__":
    main()
