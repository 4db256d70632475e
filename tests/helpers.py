"""Shared helpers of the parity tests (HIP path or simulated kernels vs. oracle / golden vectors)."""
import numpy as np
import torch

from efficientspeech_amd import CONFIGS, build_phoneme2mel, load_numpy_state_dict
from efficientspeech_amd.synth import synth_state_dict

MEL_TOL = 1e-4        # north_star: mel L-inf < 1e-4 vs the CPU reference, fp32
PRED_TOL = 2e-5       # continuous predictor outputs
MARGIN = 1e-4         # discrete decisions closer than this to a boundary may legitimately flip


def make_net(name, device, seed=1234, golden=None):
    cfg = CONFIGS[name]
    sd = synth_state_dict(cfg, seed)
    if golden is not None:              # use the exact bin edges the fixtures were generated with
        sd["encoder.pitch_decoder.pitch_bins"] = golden["pitch_bins"]
        sd["encoder.energy_decoder.energy_bins"] = golden["energy_bins"]
    net = build_phoneme2mel(cfg)
    load_numpy_state_dict(net, sd)
    return net.to(device), cfg, sd


def to_x(g_or_dict, device, train=False):
    x = {}
    for k in ("phoneme", "phoneme_mask", "pitch", "energy", "duration", "mel_len"):
        key = "in_" + k
        if key in getattr(g_or_dict, "files", g_or_dict):
            if k in ("pitch", "energy", "duration", "mel_len") and not train:
                continue
            x[k] = torch.from_numpy(np.asarray(g_or_dict[key])).to(device)
    return x


def check_against_golden(net, g, device):
    """Run one golden fixture through the module API; assert parity."""
    train = bool(g["train"])
    x = to_x(g, device, train)
    with torch.no_grad():
        enc = net.encoder._encode(x, train=train)
        out = net(x, train=train)
    if train:
        mel, mel_len, dpred = out["mel"], out["mel_len"], out["duration"]
        np.testing.assert_allclose(out["pitch"].cpu().numpy(), g["pitch"], atol=PRED_TOL, rtol=0)
        np.testing.assert_allclose(out["energy"].cpu().numpy(), g["energy"], atol=PRED_TOL, rtol=0)
        assert (out["masks"] is not None) == bool(g["has_masks"])
        if "features" in g.files:
            np.testing.assert_allclose(out["features"].cpu().numpy(), g["features"], atol=PRED_TOL, rtol=0)
        if out["masks"] is not None:
            assert np.array_equal(out["masks"][:, :, 0].cpu().numpy(), g["masks"])
    else:
        mel, mel_len, dpred = out
    np.testing.assert_allclose(dpred.cpu().numpy(), g["duration"], atol=PRED_TOL, rtol=0)
    for i, f in enumerate(enc["feats"]):
        np.testing.assert_allclose(f.cpu().numpy(), g[f"f{i}"], atol=PRED_TOL, rtol=0)
    np.testing.assert_allclose(enc["feat"].cpu().numpy(), g["feat"], atol=PRED_TOL, rtol=0)
    # discrete decisions: bit-exact
    assert np.array_equal(enc["pitch_idx"].cpu().numpy(), g["pitch_idx"])
    assert np.array_equal(enc["energy_idx"].cpu().numpy(), g["energy_idx"])
    assert np.array_equal(enc["dur"].cpu().numpy(), g["dur"])
    assert mel_len.dtype == torch.int32 and np.array_equal(mel_len.cpu().numpy(), g["mel_len"])
    assert tuple(mel.shape) == g["mel"].shape
    err = float(np.abs(mel.cpu().numpy() - g["mel"]).max())
    assert err < MEL_TOL, err
    return err


def round_margin(pred):
    """distance of each value to the nearest .5 rounding boundary"""
    frac = pred - np.floor(pred)
    return np.abs(frac - 0.5)


def bucket_margin(v, edges):
    return np.abs(v[..., None] - edges[None, :]).min(-1)


def compare_eval_with_oracle(cfg, o, enc, mel, mel_len, sd):
    """HIP (or simulated) eval outputs vs an oracle eval run on the same inputs, margin-aware for the
    discrete decisions; returns the mel L-inf error (nan if a legitimate flip changed the shapes)."""
    dp = enc["duration"].cpu().numpy()
    np.testing.assert_allclose(dp, o.duration, atol=PRED_TOL, rtol=0)
    np.testing.assert_allclose(enc["pitch"].cpu().numpy(), o.pitch, atol=PRED_TOL, rtol=0)
    np.testing.assert_allclose(enc["energy"].cpu().numpy(), o.energy, atol=PRED_TOL, rtol=0)
    dur = enc["dur"].cpu().numpy()
    flips = dur != o.dur
    assert not (flips & (round_margin(o.duration[..., 0]) > MARGIN)).any(), "duration differs away from a .5 boundary"
    pi, ei = enc["pitch_idx"].cpu().numpy(), enc["energy_idx"].cpu().numpy()
    pf, ef = pi != o.pitch_idx, ei != o.energy_idx
    assert not (pf & (bucket_margin(o.pitch[..., 0], sd["encoder.pitch_decoder.pitch_bins"]) > MARGIN)).any()
    assert not (ef & (bucket_margin(o.energy[..., 0], sd["encoder.energy_decoder.energy_bins"]) > MARGIN)).any()
    if flips.any() or pf.any() or ef.any():
        return float("nan")
    assert np.array_equal(mel_len.cpu().numpy(), o.mel_len)
    m = mel.cpu().numpy()
    assert m.shape == o.mel.shape
    err = float(np.abs(m - o.mel).max())
    assert err < MEL_TOL, err
    return err
