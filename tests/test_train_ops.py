"""Operator-level checks of the training kernels (csrc/train_ops.h, esmi_train_*) against plain PyTorch fp32 ops on the same
device, at sizes that exercise what the small reference fixture cannot: many row chunks in the two-stage reductions, weight
tiles that are not multiples of 32, strided / transposed / depthwise convolutions, several heads.  (PyTorch is the checker
here, as the numerics-test rule asks; the step-level parity is pinned to the reference fixture in test_train_step.py.)"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from efficientspeech_amd import train

gpu = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _close(a, b, rtol, what):
    a, b = a.detach(), b.detach()
    scale = max(1e-6, float(b.abs().max()))
    err = float((a - b).abs().max()) / scale
    assert err < rtol, (what, err, scale)


CONVS = [  # (c_in, c_out, k, stride, pad, groups, transposed, B, n)
    (128, 128, 1, 1, 0, 1, False, 5, 700),      # decoder pointwise: 3500 rows = 14 chunks
    (128, 80, 1, 1, 0, 1, False, 3, 401),       # mel_linear: 80 output channels (2.5 MFMA tiles)
    (128, 128, 3, 1, 1, 1, False, 4, 130),      # predictor conv
    (128, 128, 3, 2, 1, 1, False, 4, 131),      # strided merge conv, odd length
    (32, 64, 1, 1, 0, 1, False, 4, 65),
    (128, 128, 5, 1, 2, 128, False, 3, 500),    # depthwise k5
    (64, 64, 3, 1, 1, 64, False, 5, 13),        # depthwise k3, utterances shorter than two 8-row groups
    (128, 128, 5, 1, 2, 128, False, 2, 7),      # depthwise k5, utterances shorter than one group
    (128, 128, 3, 2, 0, 1, True, 4, 64),        # Fuse upsample (ConvTranspose1d, cropped to 2n)
    (128, 1, 1, 1, 0, 1, False, 4, 300),        # predictor output Linear
    (40, 24, 3, 1, 1, 1, False, 2, 50),         # channel counts the GEMM path does not take
]


@gpu
@pytest.mark.parametrize("cfg", CONVS, ids=[f"cin{c[0]}_cout{c[1]}_k{c[2]}_s{c[3]}_g{c[5]}_{'T' if c[6] else 'C'}" for c in CONVS])
@pytest.mark.parametrize("matrix_pipe", [True, False])
def test_conv_forward_and_gradients(cfg, matrix_pipe, monkeypatch):
    cin, cout, k, s, p, groups, tr, B, n = cfg
    monkeypatch.setattr(train, "USE_MATRIX_PIPE", matrix_pipe)
    x = _rand(B, n, cin, seed=1).requires_grad_()
    w = (_rand(cin, cout, k, seed=2) if tr else _rand(cout, cin // groups, k, seed=2)) * (1.0 / np.sqrt(cin * k / groups))
    w = w.detach().requires_grad_()
    b = _rand(cout, seed=3).requires_grad_()
    if tr:
        full = (n - 1) * s - 2 * p + k
        n_out = min(full, 2 * n)
        ref = F.conv_transpose1d(x.transpose(1, 2), w, b, stride=s, padding=p)[:, :, :n_out].transpose(1, 2)
    else:
        n_out = (n + 2 * p - k) // s + 1
        ref = F.conv1d(x.transpose(1, 2), w, b, stride=s, padding=p, groups=groups).transpose(1, 2)
    got = train._Conv.apply(x, w, b, s, p, groups, tr, n_out)
    _close(got, ref, 2e-5, "forward")
    dy = _rand(*ref.shape, seed=4)
    gx, gw, gb = torch.autograd.grad(got, (x, w, b), dy)
    rx, rw, rb = torch.autograd.grad(ref, (x, w, b), dy)
    _close(gx, rx, 5e-5, "dgrad")
    _close(gw, rw, 5e-5, "wgrad")
    _close(gb, rb, 5e-5, "bias grad")


@gpu
@pytest.mark.parametrize("rows,C", [(5000, 128), (300, 32), (77, 64)])
def test_layernorm(rows, C):
    x = _rand(rows, C, seed=1, scale=3.0).requires_grad_()
    g, b = (_rand(C, seed=2) + 1.0).requires_grad_(), _rand(C, seed=3).requires_grad_()
    got, ref = train._LayerNorm.apply(x, g, b), F.layer_norm(x, (C,), g, b)
    _close(got, ref, 1e-5, "forward")
    dy = _rand(rows, C, seed=4)
    for a, r, what in zip(torch.autograd.grad(got, (x, g, b), dy), torch.autograd.grad(ref, (x, g, b), dy), ("dx", "dgamma", "dbeta")):
        _close(a, r, 5e-5, what)


@gpu
@pytest.mark.parametrize("rows,C", [(300, 32), (130, 128), (40, 320)])
def test_layernorm_with_residual_and_row_mask(rows, C):
    """LN(x + res) with padded rows zeroed, in the norm's own launches (networks.py:75-76, 83-84, 301): forward, both summands'
    gradients, and the parameter gradients -- masked rows contribute nothing."""
    x, res = _rand(rows, C, seed=11, scale=2.0).requires_grad_(), _rand(rows, C, seed=12, scale=2.0).requires_grad_()
    g, b = (_rand(C, seed=13) + 1.0).requires_grad_(), _rand(C, seed=14).requires_grad_()
    mask = (torch.arange(rows, device=x.device) % 7 == 3) | (torch.arange(rows, device=x.device) >= rows - 5)
    m8 = mask.to(torch.uint8).contiguous()
    dy = _rand(rows, C, seed=15)
    for use_res, use_mask in ((True, True), (True, False), (False, True)):
        got = train._LayerNorm.apply(x, g, b, res if use_res else None, m8 if use_mask else None)
        ref = F.layer_norm(x + res if use_res else x, (C,), g, b)
        if use_mask:
            ref = ref.masked_fill(mask[:, None], 0.0)
        _close(got, ref, 1e-5, "forward")
        ins = (x, res, g, b) if use_res else (x, g, b)
        names = ("dx", "dres", "dgamma", "dbeta") if use_res else ("dx", "dgamma", "dbeta")
        for a, r, what in zip(torch.autograd.grad(got, ins, dy), torch.autograd.grad(ref, ins, dy), names):
            _close(a, r, 5e-5, what)


@gpu
def test_layernorm_with_relu_output():
    """relu(LN(relu(conv))) as the predictors run it (networks.py:152-154): the first ReLU in the convolution's launch with its backward
    in the LayerNorm's, the second in the LayerNorm's forward with dy gated by the saved output."""
    x = _rand(2, 90, 128, seed=41).requires_grad_()
    w, b = _rand(128, 128, 3, seed=42, scale=0.1).requires_grad_(), _rand(128, seed=43).requires_grad_()
    g, beta = (_rand(128, seed=44) + 1.0).requires_grad_(), _rand(128, seed=45).requires_grad_()
    m = (torch.arange(90)[None, :] >= torch.tensor([90, 31])[:, None]).to(DEV)
    for mask in (None, m.view(torch.uint8).contiguous()):
        got = train._LayerNorm.apply(train._Conv.apply(x, w, b, 1, 1, 1, False, 90, train.ACT_RELU, True), g, beta, None, mask, train.ACT_RELU, True)
        ref = F.relu(F.layer_norm(F.relu(F.conv1d(x.transpose(1, 2), w, b, padding=1).transpose(1, 2)), (128,), g, beta))
        if mask is not None:
            ref = ref.masked_fill(m[..., None], 0.0)
        _close(got, ref, 3e-5, "forward")
        dy = _rand(2, 90, 128, seed=46)
        for a, r, what in zip(torch.autograd.grad(got, (x, w, b, g, beta), dy), torch.autograd.grad(ref, (x, w, b, g, beta), dy),
                              ("dx", "dw", "db", "dgamma", "dbeta")):
            _close(a, r, 1e-4, what)


@gpu
def test_conv_activation_layernorm_chain():
    """LN(tanh(conv(x))) / LN(relu(conv(x))) with the activation in the convolution's launch and its backward in the LayerNorm's."""
    for kind, fn in ((train.ACT_TANH, torch.tanh), (train.ACT_RELU, F.relu)):
        x = _rand(2, 150, 128, seed=31).requires_grad_()
        w, b = _rand(128, 128, 1, seed=32, scale=0.15).requires_grad_(), _rand(128, seed=33).requires_grad_()
        g, beta = (_rand(128, seed=34) + 1.0).requires_grad_(), _rand(128, seed=35).requires_grad_()
        got = train._LayerNorm.apply(train._Conv.apply(x, w, b, 1, 0, 1, False, 150, kind, True), g, beta, None, None, kind)
        ref = F.layer_norm(fn(F.conv1d(x.transpose(1, 2), w, b).transpose(1, 2)), (128,), g, beta)
        _close(got, ref, 3e-5, "forward")
        dy = _rand(2, 150, 128, seed=36)
        for a, r, what in zip(torch.autograd.grad(got, (x, w, b, g, beta), dy), torch.autograd.grad(ref, (x, w, b, g, beta), dy),
                              ("dx", "dw", "db", "dgamma", "dbeta")):
            _close(a, r, 1e-4, what)


@gpu
def test_conv_with_fused_activation():
    """ReLU / tanh inside the convolution's launch (esmi_conv_desc.act): GEMM shapes, the one-channel Linear and a plain-kernel shape."""
    for (cin, cout, k, kind, fn) in ((64, 128, 3, train.ACT_RELU, F.relu), (128, 128, 1, train.ACT_TANH, torch.tanh),
                                     (128, 1, 1, train.ACT_RELU, F.relu), (6, 10, 3, train.ACT_TANH, torch.tanh)):
        x = _rand(3, 90, cin, seed=21, scale=1.0).requires_grad_()
        w, b = (_rand(cout, cin, k, seed=22, scale=0.2)).requires_grad_(), _rand(cout, seed=23).requires_grad_()
        got = train._Conv.apply(x, w, b, 1, k // 2, 1, False, 90, kind)
        ref = fn(F.conv1d(x.transpose(1, 2), w, b, padding=k // 2).transpose(1, 2))
        _close(got, ref, 2e-5, f"forward {cin}->{cout} k{k}")
        dy = _rand(3, 90, cout, seed=24)
        for a, r, what in zip(torch.autograd.grad(got, (x, w, b), dy), torch.autograd.grad(ref, (x, w, b), dy), ("dx", "dw", "db")):
            _close(a, r, 1e-4, f"{what} {cin}->{cout} k{k}")


@gpu
@pytest.mark.parametrize("kind,fn", [(train.ACT_RELU, F.relu), (train.ACT_GELU, F.gelu), (train.ACT_TANH, torch.tanh)])
def test_activations(kind, fn):
    x = _rand(3, 1000, 64, seed=5, scale=2.0).requires_grad_()
    got, ref = train._Act.apply(x, kind), fn(x)
    _close(got, ref, 2e-6, "forward")
    dy = _rand(3, 1000, 64, seed=6)
    _close(torch.autograd.grad(got, x, dy)[0], torch.autograd.grad(ref, x, dy)[0], 1e-5, "backward")


@gpu
@pytest.mark.parametrize("B,N,C,h", [(3, 128, 32, 1), (2, 70, 64, 2), (2, 33, 128, 4)])
def test_attention_core(B, N, C, h):
    qkv = _rand(B, N, 3 * h * C, seed=7, scale=0.5).requires_grad_()
    got = train._AttnCore.apply(qkv, h)
    q, k, v = qkv.reshape(B, N, 3, h, C).permute(2, 0, 3, 1, 4).unbind(0)          # blocks.py:46-48
    attn = ((q @ k.transpose(-2, -1)) * (C // h) ** -0.5).softmax(dim=-1)
    ref = (attn @ v).transpose(1, 2).reshape(B, N, -1)
    _close(got, ref, 1e-5, "forward")
    dy = _rand(B, N, h * C, seed=8)
    _close(torch.autograd.grad(got, qkv, dy)[0], torch.autograd.grad(ref, qkv, dy)[0], 5e-5, "backward")


@gpu
def test_embedding_repeat_cat_mask_add():
    table = _rand(153, 128, seed=9).requires_grad_()
    ids = torch.randint(0, 153, (6, 200), generator=torch.Generator().manual_seed(1)).to(DEV)
    got, ref = train._Embedding.apply(ids, table, 0), F.embedding(ids, table, padding_idx=0)
    assert torch.equal(got, ref)
    dy = _rand(6, 200, 128, seed=10)
    _close(torch.autograd.grad(got, table, dy)[0], torch.autograd.grad(ref, table, dy)[0], 2e-5, "embedding grad")
    # length regulator
    feat = _rand(3, 50, 16, seed=11).requires_grad_()
    dur = torch.randint(0, 7, (3, 50), generator=torch.Generator().manual_seed(2)).to(DEV)
    cum = torch.cumsum(dur, 1).to(torch.int32).contiguous()
    L = int(cum[:, -1].max()) + 5
    got = train._Repeat.apply(feat, cum, L)
    ref = torch.stack([F.pad(f.repeat_interleave(d, dim=0), (0, 0, 0, L - int(d.sum()))) for f, d in zip(feat, dur)])
    assert torch.equal(got, ref)
    dy = _rand(3, L, 16, seed=12)
    _close(torch.autograd.grad(got, feat, dy)[0], torch.autograd.grad(ref, feat, dy)[0], 1e-5, "repeat grad")
    # cat / mask / add
    a, b = _rand(4, 30, 8, seed=13).requires_grad_(), _rand(4, 30, 24, seed=14).requires_grad_()
    m = (torch.arange(30)[None, :] >= torch.tensor([30, 12, 1, 29])[:, None]).to(DEV)
    m8 = m.view(torch.uint8) if m.dtype != torch.uint8 else m
    got = train._MaskRows.apply(train._Cat.apply(None, 0, a, train._Add.apply(b, b)), m8)
    ref = torch.cat([a, b + b], -1).masked_fill(m[..., None], 0)
    assert torch.equal(got, ref)
    dy = _rand(4, 30, 32, seed=15)
    for x_, y_ in zip(torch.autograd.grad(got, (a, b), dy), torch.autograd.grad(ref, (a, b), dy)):
        assert torch.equal(x_, y_)
    # the cat that masks some of its parts itself (networks.py:366-368: three of the four parts are masked_fill'ed first)
    c = _rand(4, 30, 5, seed=16).requires_grad_()
    got = train._Cat.apply(m8, 0b110, a, b, c)
    ref = torch.cat([a, b.masked_fill(m[..., None], 0), c.masked_fill(m[..., None], 0)], -1)
    assert torch.equal(got, ref)
    dy = _rand(4, 30, 37, seed=17)
    for x_, y_ in zip(torch.autograd.grad(got, (a, b, c), dy), torch.autograd.grad(ref, (a, b, c), dy)):
        assert torch.equal(x_, y_)


@gpu
def test_loss_and_adamw_against_torch():
    B, T, L, nm = 5, 40, 300, 80
    mel_p, mel = _rand(B, L, nm, seed=1).requires_grad_(), _rand(B, L, nm, seed=2)
    pp, p_ = _rand(B, T, seed=3).requires_grad_(), _rand(B, T, seed=4)
    ep, e_ = _rand(B, T, seed=5).requires_grad_(), _rand(B, T, seed=6)
    dp = _rand(B, T, seed=7).abs().requires_grad_()
    d_ = torch.randint(0, 9, (B, T), generator=torch.Generator().manual_seed(3)).to(DEV).to(torch.int32)
    mm = (torch.arange(L)[None, :] >= torch.tensor([300, 250, 17, 299, 1])[:, None]).to(DEV)
    pm = (torch.arange(T)[None, :] >= torch.tensor([40, 33, 2, 39, 1])[:, None]).to(DEV)
    parts, tot = train._Loss.apply(mel_p, pp, ep, dp, mel, p_, e_, d_, mm.view(torch.uint8), pm.view(torch.uint8))
    got = list(parts) + [tot]
    sel, ps = ~mm[..., None], ~pm
    ref = [F.l1_loss(mel_p.masked_select(sel), mel.masked_select(sel)), F.mse_loss(pp.masked_select(ps), p_.masked_select(ps)),
           F.mse_loss(ep.masked_select(ps), e_.masked_select(ps)),
           F.mse_loss(torch.log(dp.masked_select(ps) + 1), torch.log(d_.masked_select(ps).float() + 1))]
    total = 10 * ref[0] + 2 * ref[1] + 2 * ref[2] + ref[3]
    for a, r in zip(got[:4], ref):
        assert abs(float(a.detach()) - float(r.detach())) < 2e-5 * abs(float(r.detach()))
    assert abs(float(got[4].detach()) - float(total.detach())) < 2e-5 * abs(float(total.detach()))
    for a, r, what in zip(torch.autograd.grad(got[4], (mel_p, pp, ep, dp)), torch.autograd.grad(total, (mel_p, pp, ep, dp)),
                          ("d mel", "d pitch", "d energy", "d duration")):
        _close(a, r, 2e-5, what)
    # AdamW: five steps of the kernel against torch.optim.AdamW on the same gradients
    from efficientspeech_amd import _lib
    from efficientspeech_amd.networks import _runtime, _ptr
    n = 100_003
    p0 = _rand(n, seed=20)
    ref_p = p0.clone().requires_grad_()
    opt = torch.optim.AdamW([ref_p], lr=1e-3, weight_decay=1e-2)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    lib, st = _runtime(p)
    for t in range(1, 6):
        g = _rand(n, seed=30 + t)
        ref_p.grad = g.clone()
        opt.step()
        lib.esmi_train_adamw_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, t, 1.0, st)
    assert float((p - ref_p.detach()).abs().max()) < 2e-6


@gpu
@pytest.mark.parametrize("name", ["small", "base"])
def test_training_runs_on_the_wider_configs(name):
    """small (3 blocks, reduction 2) and base (2 / 4 heads, expansion 2, k = 5, depth 3): the loss falls and stays finite."""
    from efficientspeech_amd import CONFIGS, build_phoneme2mel
    from efficientspeech_amd.synth import synth_state_dict
    cfg = CONFIGS[name]
    net = build_phoneme2mel(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 1234).items()})
    net = net.to(DEV).train()
    x, y = train.synthetic_batch(4, 24, 3, DEV)
    step = train.TrainStep(net, lr=1e-3)
    first = float(step.step(x, y)[4])
    for _ in range(8):
        last = float(step.step(x, y)[4])
    assert np.isfinite(last) and last < first, (first, last)


# ---------------------------------------------------------------------- the same operator checks on the CPU wave simulator
# (small shapes: a fiber per GPU thread), so the GPU-less tier also pins every training kernel to PyTorch's own op
SIM_CONVS = [(32, 32, 3, 1, 1, 1, False, 2, 37), (32, 64, 3, 2, 1, 1, False, 2, 19), (64, 32, 3, 2, 0, 1, True, 2, 9),
             (32, 32, 5, 1, 2, 32, False, 2, 300), (32, 1, 1, 1, 0, 1, False, 2, 21), (40, 24, 3, 1, 1, 1, False, 1, 11),
             (128, 80, 1, 1, 0, 1, False, 1, 300)]


def _sim_case(fn):
    """Run a GPU-marked check body on host tensors through the simulator build."""
    import tests.test_train_ops as me
    from tests.simlib import use_sim
    old = me.DEV
    me.DEV = "cpu"
    try:
        with use_sim():
            fn()
    finally:
        me.DEV = old


_nogpu = pytest.mark.filterwarnings("ignore")


@pytest.mark.parametrize("cfg", SIM_CONVS, ids=[f"cin{c[0]}_cout{c[1]}_k{c[2]}_s{c[3]}_g{c[5]}_{'T' if c[6] else 'C'}" for c in SIM_CONVS])
def test_simulated_conv_forward_and_gradients(cfg, monkeypatch):
    _sim_case(lambda: test_conv_forward_and_gradients(cfg, True, monkeypatch))


def test_simulated_layernorm_attention_loss():
    _sim_case(lambda: (test_layernorm(300, 32), test_layernorm(77, 64), test_layernorm_with_residual_and_row_mask(130, 128),
                       test_layernorm_with_residual_and_row_mask(40, 320), test_conv_with_fused_activation(),
                       test_conv_activation_layernorm_chain(), test_layernorm_with_relu_output(),
                       test_attention_core(2, 33, 32, 2), test_activations(train.ACT_GELU, F.gelu),
                       test_embedding_repeat_cat_mask_add(), test_loss_and_adamw_against_torch()))
