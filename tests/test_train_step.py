"""Training step against the reference-generated fixture tests/golden/tiny_train_step.npz
(tools/gen_golden_train.py: the reference's train=True forward, model.py's loss arithmetic, torch autograd, torch AdamW)."""
import os

import numpy as np
import pytest
import torch

from efficientspeech_amd import CONFIGS, build_phoneme2mel
from efficientspeech_amd.synth import synth_state_dict

GOLD = os.path.join(os.path.dirname(__file__), "golden", "tiny_train_step.npz")
pytestmark = pytest.mark.gpu


def _setup(dev):
    from efficientspeech_amd import train
    g = np.load(GOLD)
    cfg = CONFIGS["tiny"]
    net = build_phoneme2mel(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 1234).items()}, strict=True)
    net = net.to(dev).train()
    t = lambda k: torch.from_numpy(g["in_" + k]).to(dev)     # noqa: E731
    x = {"phoneme": t("phoneme"), "phoneme_mask": t("phoneme_mask"), "pitch": t("pitch"), "energy": t("energy"),
         "duration": t("duration"), "mel_len": t("mel_len"), "mel_mask": t("mel_mask")}
    y = {"mel": t("mel")}
    return train, g, net, x, y


def test_loss_and_gradients_match_reference():
    train, g, net, x, y = _setup("cuda")
    step = train.TrainStep(net)
    step.flat.zero_grad()
    losses = train.training_loss(net, x, y)
    losses[4].backward()
    got = losses.detach().cpu().numpy().astype(np.float64)
    assert np.allclose(got[:4], g["losses"], rtol=2e-5, atol=1e-6), (got, g["losses"])
    assert abs(got[4] - float(g["total"])) < 2e-5 * abs(float(g["total"]))
    named = dict(net.named_parameters())
    worst = 0.0
    for k in g.files:
        if not k.startswith("grad."):
            continue
        ref = g[k]
        mine = named[k[5:]].grad.detach().cpu().numpy()
        scale = max(1e-6, float(np.abs(ref).max()))
        err = float(np.abs(mine - ref).max()) / scale
        worst = max(worst, err)
        assert err < 2e-4, (k, err, scale)
    for k in g["no_grad_params"]:
        assert not any(str(k) == n for n in step.flat.names), k          # left out of the optimizer, as torch's grad-is-None rule
    assert worst > 0.0


def test_adamw_step_matches_reference():
    train, g, net, x, y = _setup("cuda")
    step = train.TrainStep(net, lr=1e-3, weight_decay=1e-6)
    losses = step.step(x, y)
    assert abs(float(losses[4]) - float(g["total"])) < 2e-5 * abs(float(g["total"]))
    named = dict(net.named_parameters())
    for k in g.files:
        if k.startswith("after."):
            ref, mine = g[k], named[k[6:]].detach().cpu().numpy()
            assert np.abs(mine - ref).max() < 2e-6, (k, float(np.abs(mine - ref).max()))


def test_training_reduces_the_loss_and_inference_sees_the_update():
    train, g, net, x, y = _setup("cuda")
    step = train.TrainStep(net, lr=1e-3)
    first = float(step.step(x, y)[4])
    for _ in range(20):
        last = float(step.step(x, y)[4])
    assert last < 0.9 * first, (first, last)
    net.eval()
    with torch.no_grad():
        mel, mel_len, _ = net({"phoneme": x["phoneme"], "phoneme_mask": x["phoneme_mask"]})
    assert bool(torch.isfinite(mel).all())
