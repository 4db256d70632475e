"""Pins the CPU oracle (oracle/es_oracle.c) to outputs of the reference itself.

The fixtures under tests/golden/ were produced by tools/gen_golden.py running the reference
modules (/root/reference/layers) in the build container; this test never touches the reference.
Tolerances: continuous tensors 2e-5 (fp32 reference vs double-accumulating oracle), discrete
decisions (bucket indices, rounded durations, mel_len, length-regulator rows) bit-exact.
"""
import glob
import os
import zlib

import numpy as np
import pytest

from efficientspeech_amd.config import CONFIGS
from efficientspeech_amd.synth import synth_state_dict
from oracle import oracle

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                if os.path.basename(p).startswith(("tiny_", "small_", "base_")) and "_train_step" not in os.path.basename(p))   # forward
#                              fixtures of the acoustic model (the vocoder, the training step and the loader have their own tests)
TOL = 2e-5


def _crc(sd):
    c = 0
    for k, v in sd.items():
        c = zlib.crc32(v.tobytes(), zlib.crc32(k.encode(), c))
    return c


@pytest.fixture(scope="module")
def weights():
    cache = {}

    def get(name, g):
        if name not in cache:
            sd = synth_state_dict(CONFIGS[name], int(g["seed"]))
            sd["encoder.pitch_decoder.pitch_bins"] = g["pitch_bins"]
            sd["encoder.energy_decoder.energy_bins"] = g["energy_bins"]
            assert _crc(sd) == int(g["weights_crc"]), "synthetic weight recipe drifted from the fixtures"
            cache[name] = oracle.Weights(sd)
        return cache[name]
    return get


def test_fixtures_present():
    assert len(GOLDEN) == 18


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_reference(path, weights):
    g = np.load(path)
    name = os.path.basename(path).split("_")[0]
    cfg = CONFIGS[name]
    w = weights(name, g)
    train = bool(g["train"])
    kw = {}
    if train:
        kw = dict(pitch=g["in_pitch"], energy=g["in_energy"], duration=g["in_duration"],
                  max_mel_len=int(g["in_mel_len"].max()))
    mask = g["in_phoneme_mask"] if "in_phoneme_mask" in g.files else None
    o = oracle.phoneme2mel(cfg, w, g["in_phoneme"], mask, taps=True, **kw)
    # stage taps (Encoder blocks, Fuse)
    for i in range(cfg.depth):
        np.testing.assert_allclose(o.f_taps[i], g[f"f{i}"], atol=TOL, rtol=0)
    np.testing.assert_allclose(o.fused, g["fused"], atol=TOL, rtol=0)
    # predictions (AcousticDecoder x3)
    np.testing.assert_allclose(o.duration, g["duration"], atol=TOL, rtol=0)
    if train:
        np.testing.assert_allclose(o.pitch, g["pitch"], atol=TOL, rtol=0)
        np.testing.assert_allclose(o.energy, g["energy"], atol=TOL, rtol=0)
    # discrete decisions: bit-exact
    assert np.array_equal(o.pitch_idx, g["pitch_idx"])
    assert np.array_equal(o.energy_idx, g["energy_idx"])
    assert np.array_equal(o.dur, g["dur"])
    assert np.array_equal(o.mel_len, g["mel_len"])
    # variance-adaptor concat and length regulator
    np.testing.assert_allclose(o.feat, g["feat"], atol=TOL, rtol=0)
    if "features" in g.files:
        np.testing.assert_allclose(o.features, g["features"], atol=TOL, rtol=0)
    B = g["in_phoneme"].shape[0]
    if B > 1:
        assert np.array_equal(o.masks, g["masks"])
    else:
        assert o.masks is None
    # mel
    assert o.mel.shape == g["mel"].shape
    err = np.abs(o.mel - g["mel"]).max()
    assert err < 1e-4, err
    assert err < 5e-5, f"oracle drifted further from the reference than expected: {err}"


def test_length_regulator_rule():
    """cumsum rule of networks.py:233-244 / acoustic.py:33-42 incl. zeros, crop and pad."""
    dur = np.array([[2, 0, 3, 1], [0, 0, 0, 0], [1, 1, 1, 5]], np.int32)
    idx = oracle.length_regulate(dur, 7)
    assert idx.tolist() == [[0, 0, 2, 2, 2, 3, -1], [-1] * 7, [0, 1, 2, 3, 3, 3, 3]]


def test_mask_from_lengths():
    m = oracle.mask_from_lengths([3, 0, 5], 5)
    assert m.tolist() == [[False] * 3 + [True] * 2, [True] * 5, [False] * 5]


# ---------------------------------------------------------------- HiFi-GAN generator
HIFIGAN_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "hifigan_*.npz")))


@pytest.mark.parametrize("path", HIFIGAN_GOLDEN, ids=[os.path.basename(p)[:-4] for p in HIFIGAN_GOLDEN])
def test_hifigan_oracle_matches_reference_vectors(path):
    """oracle.hifigan (oracle/es_oracle.c eso_hifigan) vs the waveform the reference's hifigan.Generator produced
    (tools/gen_golden_hifigan.py) on the same seeded weights and mel."""
    from efficientspeech_amd.hifigan import HIFIGAN_CONFIGS, synth_hifigan_state_dict
    g = np.load(path)
    h = HIFIGAN_CONFIGS[str(g["config"])]
    w = oracle.Weights(synth_hifigan_state_dict(h, 1234))
    wav = oracle.hifigan(h, w, g["mel"])
    assert wav.shape == g["wav"].shape
    assert np.abs(wav - g["wav"]).max() < 2e-5, np.abs(wav - g["wav"]).max()


def test_hifigan_weight_norm_folding_and_key_table():
    """fold_weight_norm(g, v) == the plain weight the fixture was generated with; the module mirror's keys/shapes are the
    spec table (which gen_golden_hifigan.py asserted against the reference's Generator)."""
    import torch
    from efficientspeech_amd.hifigan import (HIFIGAN_CONFIGS, Generator, fold_weight_norm, hifigan_state_dict_spec,
                                             synth_hifigan_state_dict)
    g = np.load([p for p in HIFIGAN_GOLDEN if "hifigan_v2" in p][0])
    sd = synth_hifigan_state_dict(HIFIGAN_CONFIGS["v2"], 1234)
    wn = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("wn.")}
    folded = fold_weight_norm(wn)
    assert sorted(folded) == ["conv_post.bias", "conv_post.weight", "ups.3.bias", "ups.3.weight"]
    for k, v in folded.items():
        np.testing.assert_allclose(v.numpy(), sd[k], rtol=2e-6, atol=1e-7)
    for name, h in HIFIGAN_CONFIGS.items():
        net = Generator(h)
        assert sorted((k, tuple(v.shape)) for k, v in net.state_dict().items()) == sorted(hifigan_state_dict_spec(h))
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_hifigan_state_dict(h, 3).items()}, strict=True)
    assert sum(p.numel() for p in Generator(HIFIGAN_CONFIGS["v2"]).parameters()) == 925_985   # = the reference LJ_V2/generator_v2 checkpoint (weight_v + bias tensors), "926k"


def test_oracle_training_loss_and_adamw_match_reference_fixture():
    """the oracle's restatement of model.py:167-216 (loss) and of torch.optim.AdamW's update, pinned to the
    reference-generated training fixture (tools/gen_golden_train.py)."""
    from efficientspeech_amd.synth import synth_state_dict
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tiny_train_step.npz"))
    losses, grads = oracle.training_loss(g["mel_pred"], g["in_mel"], g["in_mel_mask"], g["pitch_pred"], g["in_pitch"], g["energy_pred"],
                                         g["in_energy"], g["duration_pred"], g["in_duration"], g["in_phoneme_mask"])
    assert np.allclose(losses[:4], g["losses"], rtol=1e-6) and abs(losses[4] - float(g["total"])) < 1e-6 * float(g["total"])
    assert grads[0].shape == g["mel_pred"].shape and np.all(grads[0][g["in_mel_mask"]] == 0)
    sd = synth_state_dict(CONFIGS["tiny"], 1234)
    n = 0
    for k in g.files:
        if k.startswith("after."):
            name = k[6:]
            p1, _, _ = oracle.adamw_step(sd[name], g["grad." + name], np.zeros_like(sd[name]), np.zeros_like(sd[name]), 1)
            assert np.abs(p1 - g[k]).max() < 1e-7, name
            n += 1
    assert n >= 5


def test_torch_mirror_eval_forward_matches_oracle():
    """tests/torch_mirror.py (the stock-PyTorch restatement bench.py times as `pytorch_rocm_ops`) against the C oracle: eval flow
    with injected durations, padded batch."""
    import torch
    from efficientspeech_amd import build_phoneme2mel
    from efficientspeech_amd.synth import synth_state_dict, synth_phonemes
    from tests import torch_mirror as M
    for name in ("tiny", "small"):
        cfg = CONFIGS[name]
        sd = synth_state_dict(cfg, 1234)
        net = build_phoneme2mel(cfg)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        ids, mask = synth_phonemes(3, 21, 5, [21, 13, 7])
        dur = np.full((3, 21), 4, np.int32)
        mel, mel_len, _ = M.eval_forward(net, {"phoneme": torch.from_numpy(ids), "phoneme_mask": torch.from_numpy(mask),
                                               "duration_forced": torch.from_numpy(dur)})
        o = oracle.phoneme2mel(cfg, oracle.Weights(sd), ids, mask, pitch=None, energy=None, duration=dur)
        assert np.array_equal(mel_len.numpy(), o.mel_len) and np.abs(mel.numpy() - o.mel).max() < 2e-5
