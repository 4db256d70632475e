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


def check_attention_sizes(net, cfg, sd, device, sizes=((2, 100), (1, 200), (3, 128))):
    """SelfAttention.forward at lengths with 3 .. 8 key tiles: the (utterance, head) workgroup kernel that stages K and V in LDS
    (csrc/attention.h attn_lds_kernel; on the GPU it needs >= 128 heads in flight, in the simulator build it always runs)."""
    from oracle import oracle
    rng = np.random.default_rng(11)
    enc = net.encoder.encoder
    for i, blk in enumerate(enc.attn_blocks):
        pre = f"encoder.encoder.attn_blocks.{i}."
        for B, N in sizes:
            x = rng.standard_normal((B, N, enc.dim_outs[i])).astype(np.float32)
            ref = oracle.self_attention(x, sd[pre + "2.qkv.weight"], sd[pre + "2.proj.weight"], sd[pre + "2.proj.bias"], enc.heads[i])
            with torch.no_grad():
                y, _ = blk[2](torch.from_numpy(x).to(device))
            np.testing.assert_allclose(y.cpu().numpy(), ref, atol=PRED_TOL, rtol=0)


def check_lds_gemm_edges(device, cases):
    """MixFFN.forward = three launches of the per-op plan's LDS-staged GEMM (convgemm_dma_kernel; a k = 3 convolution in the middle)
    against a torch fp64 restatement (blocks.py:22-29) on shapes at the kernel's edges: one-position utterances, utterances shorter
    than the taps' reach, row counts off the 128 / 256-row workgroup tile, channel counts off the 128-column tile."""
    import torch.nn.functional as F
    from efficientspeech_amd.networks import MixFFN
    torch.manual_seed(0)
    for B, N, Cc, E in cases:
        m = MixFFN(Cc, E).to(device)
        x = torch.randn(B, N, Cc).to(device)
        with torch.no_grad():
            y = m(x)
            h = F.linear(x.double(), m.mlp1.weight.double(), m.mlp1.bias.double())
            h = F.gelu(F.conv1d(h.transpose(1, 2), m.conv.weight.double(), m.conv.bias.double(), padding=1).transpose(1, 2))
            ref = F.linear(h, m.mlp2.weight.double(), m.mlp2.bias.double())
        err = float((y.double() - ref).abs().max()) / float(ref.abs().max())
        assert torch.isfinite(y).all() and err < 5e-6, (B, N, Cc, E, err)


def check_submodule_forwards(net, cfg, sd, device, seed=3):
    """The reference's sub-modules called on their own (SelfAttention / MixFFN / AcousticDecoder.forward, get_embedding,
    blocks.py:22-29,43-71, networks.py:128-165) against the oracle's restatement of the same functions."""
    from oracle import oracle
    rng = np.random.default_rng(seed)
    enc = net.encoder.encoder
    for i, blk in enumerate(enc.attn_blocks):
        attn, ffn = blk[2], blk[3]
        Cc, B, N = enc.dim_outs[i], 2, 37
        x = rng.standard_normal((B, N, Cc)).astype(np.float32)
        pre = f"encoder.encoder.attn_blocks.{i}."
        ref = oracle.self_attention(x, sd[pre + "2.qkv.weight"], sd[pre + "2.proj.weight"], sd[pre + "2.proj.bias"], enc.heads[i])
        mask = np.arange(2 * N)[None, :] >= np.array([2 * N, 41])[:, None]
        with torch.no_grad():
            y, am = attn(torch.from_numpy(x).to(device), mask=torch.from_numpy(mask).to(device), pool=2)
            y2, am2 = attn(torch.from_numpy(x).to(device))
        assert am2 is None and torch.equal(y, y2)
        np.testing.assert_allclose(y.cpu().numpy(), ref, atol=PRED_TOL, rtol=0)
        pooled = mask.reshape(B, N, 2).max(-1)
        assert am.shape == (B, N, Cc) and am.dtype == torch.bool and np.array_equal(am[:, :, 0].cpu().numpy(), pooled)
        ref = oracle.mixffn(x, sd[pre + "3.mlp1.weight"], sd[pre + "3.mlp1.bias"], sd[pre + "3.conv.weight"],
                            sd[pre + "3.conv.bias"], sd[pre + "3.mlp2.weight"], sd[pre + "3.mlp2.bias"])
        with torch.no_grad():
            y = ffn(torch.from_numpy(x).to(device))
        np.testing.assert_allclose(y.cpu().numpy(), ref, atol=PRED_TOL, rtol=0)
    w = oracle.Weights(sd)
    fused = rng.standard_normal((3, 29, cfg.dim)).astype(np.float32)
    for which, dec in enumerate((net.encoder.pitch_decoder, net.encoder.energy_decoder, net.encoder.duration_decoder)):
        pred, feats = oracle.acoustic(cfg, w, which, fused)
        with torch.no_grad():
            out = dec(torch.from_numpy(fused).to(device))
        if which == 2:
            np.testing.assert_allclose(out[0].cpu().numpy(), pred, atol=PRED_TOL, rtol=0)
            np.testing.assert_allclose(out[1].cpu().numpy(), feats, atol=PRED_TOL, rtol=0)
        else:
            assert out.shape == (3, 29, 1)
            np.testing.assert_allclose(out.cpu().numpy(), pred, atol=PRED_TOL, rtol=0)
            bins = sd[f"encoder.{'pitch' if which == 0 else 'energy'}_decoder.{'pitch' if which == 0 else 'energy'}_bins"]
            emb = sd[f"encoder.{'pitch' if which == 0 else 'energy'}_decoder.{'pitch' if which == 0 else 'energy'}_embedding.weight"]
            v = rng.uniform(bins[0] - 1, bins[-1] + 1, size=(3, 29)).astype(np.float32)
            v[0, :len(bins)] = bins[:29][:len(bins)] if len(bins) >= 29 else v[0, :len(bins)]   # values exactly ON edges: right=False
            with torch.no_grad():
                e_t = dec.get_embedding(None, torch.from_numpy(v).to(device), None)
                e_p = dec.get_embedding(torch.from_numpy(v[..., None]).to(device), None, None)
            idx = np.searchsorted(bins, v, side="left")
            assert e_t.shape == (3, 29, cfg.dim) and e_p.shape == (3, 29, 1, cfg.dim)
            assert np.array_equal(e_t.cpu().numpy(), emb[idx]) and np.array_equal(e_p.cpu().numpy()[:, :, 0], emb[idx])
    assert net.encoder.duration_decoder.get_embedding(None, None, None) is None


def check_decoder_head(net, cfg, device, seed=11):
    """MelDecoder's first stage at PHONEME rate (esmi_decoder_head_f32: GEMM + tanh + LayerNorm in one launch) against plain torch
    fp32 on the same weights, and the decoder fed with it (h0) against the decoder running the stage itself at frame rate."""
    import ctypes as C
    from efficientspeech_amd import networks
    dec = net.decoder
    rng = np.random.default_rng(seed)
    B, T, D = 3, 21, 4
    feat = torch.from_numpy(rng.standard_normal((B, T, cfg.d4)).astype(np.float32)).to(device)
    lib, stream = networks._runtime(feat)
    head = dec._head(lib, stream)
    assert head is not None and head[0].proj_w
    h0 = torch.empty((B, T, cfg.dx2), dtype=torch.float32, device=device)
    lib.esmi_decoder_head_f32(C.byref(head[0]), B * T, feat.data_ptr(), h0.data_ptr(), stream)
    with torch.no_grad():
        ref = torch.nn.functional.layer_norm(torch.tanh(torch.nn.functional.linear(feat.cpu(), dec.proj[0].weight.cpu(), dec.proj[0].bias.cpu())),
                                             (cfg.dx2,), dec.proj[2].weight.cpu(), dec.proj[2].bias.cpu(), 1e-5)
        err = float((h0.cpu() - ref).abs().max())
        assert err < 2e-5, err
        # the decoder gathering h0 == the decoder computing its first stage per frame (ragged lengths: padding frames, the final mask)
        dur = torch.from_numpy(rng.integers(0, D + 3, size=(B, T)).astype(np.int32)).to(device)
        dur[1, 15:] = 0
        cum = torch.cumsum(dur, 1).to(torch.int32).contiguous()
        mel_len = cum[:, -1].contiguous()
        L = int(mel_len.max())
        a = dec._fused(feat, cum, mel_len, None, L, True, L, h0=h0)
        b = dec._fused(feat, cum, mel_len, None, L, True, L, h0=None)
        d = float((a - b).abs().max())
        assert d < 2e-5, d


def check_embedding_folded_into_merge_conv(net, cfg, device, seed=23):
    """Block 0 with the embedding folded into the composed merge conv (esmi_encoder_block_weights.emb_conv: one table per tap) against
    the same block running the gather + contraction: ragged lengths, ids at both ends of the vocabulary, the padding id, a sequence of
    one position; the table itself against its fp64 definition."""
    import os
    enc = net.encoder.encoder
    rng = np.random.default_rng(seed)
    blocks, _ = enc._packed(*__import__("efficientspeech_amd").networks._runtime(enc.embed.weight))
    assert blocks[0][0].emb_conv and not blocks[1][0].emb_conv
    merge, merge1 = enc.attn_blocks[0][0], enc.attn_blocks[0][1]
    with torch.no_grad():
        comp = torch.einsum("om,mij->joi", merge1.weight[:, :, 0].double(), merge.weight.double())          # W'[j] (Cout, Cin)
        ref = torch.einsum("vi,joi->jvo", enc.embed.weight.double(), comp).float().cpu().numpy()
    tab = [t for t in blocks[0][1] if t.dim() == 3 and tuple(t.shape) == ref.shape]
    assert len(tab) == 1 and np.abs(tab[0].cpu().numpy() - ref).max() < 1e-6 * max(1.0, np.abs(ref).max())
    outs = {}
    for fold in ("1", "0"):
        os.environ["ESMI_FOLD_EMBED"] = fold
        enc._cache.invalidate()
        try:
            res = []
            for B, T in ((3, 37), (2, 128), (1, 1)):
                ids = rng.integers(1, 153, size=(B, T)) if fold == "1" else outs[("ids", B, T)]
                if fold == "1":
                    ids[0, 0], ids[-1, -1] = 152, 1
                    if T > 8:
                        ids[-1, T - 5:] = 0                      # padding id (row 0 of the table)
                    outs[("ids", B, T)] = ids
                mask = torch.from_numpy(ids == 0).to(device) if B > 1 else None
                with torch.no_grad():
                    feats, _ = enc(torch.from_numpy(ids).to(device), mask)
                res.append([f.cpu().numpy() for f in feats])
            outs[fold] = res
        finally:
            os.environ.pop("ESMI_FOLD_EMBED", None)
            enc._cache.invalidate()
    for a, b in zip(outs["1"], outs["0"]):
        for fa, fb in zip(a, b):
            assert np.abs(fa - fb).max() < 2e-5, np.abs(fa - fb).max()


def check_ffn_linear_folded_into_conv(net, cfg, device, seed=29):
    """MixFFN with its Linear folded into the k = 3 conv (esmi_encoder_block_weights.ffn_cw, position-dependent bias at the ends of a
    sequence) against the three contractions of the reference (ESMI_FOLD_FFN=0: no folded weights -> one kernel per op): padded batch
    (the masked rows carry the Linear's bias into the conv), odd lengths, sequences of one and two positions; and the folded tensors
    against their fp64 definitions."""
    import os
    from efficientspeech_amd import networks
    enc = net.encoder.encoder
    rng = np.random.default_rng(seed)
    blocks, _ = enc._packed(*networks._runtime(enc.embed.weight))
    for (wts, keep), blk in zip(blocks, enc.attn_blocks):
        assert wts.ffn_cw and wts.ffn_cwp and wts.ffn_cb and wts.ffn_cb_first and wts.ffn_cb_last
        ffn = blk[3]
        with torch.no_grad():
            cw = ffn.conv.weight.double().permute(2, 0, 1)                      # (3, out, in)
            ref_w = torch.einsum("jom,mi->joi", cw, ffn.mlp1.weight.double()).float().cpu().numpy()
            e = torch.einsum("jom,m->jo", cw, ffn.mlp1.bias.double())
            ref_b = (ffn.conv.bias.double() + e.sum(0)).float().cpu().numpy()
        got_w = [t for t in keep if t.dim() == 3 and tuple(t.shape) == ref_w.shape and t.data_ptr() == wts.ffn_cw]
        assert len(got_w) == 1 and np.abs(got_w[0].cpu().numpy() - ref_w).max() < 1e-6 * max(1.0, np.abs(ref_w).max())
        got_b = [t for t in keep if t.dim() == 1 and t.data_ptr() == wts.ffn_cb]
        assert len(got_b) == 1 and np.abs(got_b[0].cpu().numpy() - ref_b).max() < 1e-6 * max(1.0, np.abs(ref_b).max())
    cases = [(3, 37, [37, 20, 5]), (2, 128, [128, 77]), (1, 1, [1]), (2, 2, [2, 1])]
    ids_all = []
    for B, T, lens in cases:
        ids = rng.integers(1, 153, size=(B, T))
        for b, n in enumerate(lens):
            ids[b, n:] = 0
        ids_all.append(ids)
    outs = {}
    for fold in ("1", "0"):
        os.environ["ESMI_FOLD_FFN"] = fold
        enc._cache.invalidate()
        try:
            res = []
            for ids in ids_all:
                mask = torch.from_numpy(ids == 0).to(device) if ids.shape[0] > 1 else None
                with torch.no_grad():
                    feats, _ = enc(torch.from_numpy(ids).to(device), mask)
                res.append([f.cpu().numpy() for f in feats])
            outs[fold] = res
        finally:
            os.environ.pop("ESMI_FOLD_FFN", None)
            enc._cache.invalidate()
    for a, b in zip(outs["1"], outs["0"]):
        for fa, fb in zip(a, b):
            assert np.abs(fa - fb).max() < 3e-5, np.abs(fa - fb).max()


def check_decoder_chunk_walk(net, cfg, device, cases=((2, 40, 9), (5, 64, 7)), seed=17):
    """dx2 = 256: the decoder walking an utterance chunk by chunk with carried rows (a workspace) against the same kernel
    recomputing both halos of every 128-frame window (no workspace): identical rows, so the outputs must agree to rounding --
    ragged lengths, several segments per utterance (small B) and, on the GPU, whole-utterance walks (B >= 256)."""
    import ctypes as C
    from efficientspeech_amd import networks
    dec = net.decoder
    rng = np.random.default_rng(seed)
    for B, T, D in cases:
        feat = torch.from_numpy(rng.standard_normal((B, T, cfg.d4)).astype(np.float32)).to(device)
        dur = torch.from_numpy(rng.integers(1, D + 1, size=(B, T)).astype(np.int32)).to(device)
        if B > 1:
            dur[1, T // 2:] = 0                                   # a short utterance: padding frames, all-padding chunks
        cum = torch.cumsum(dur, 1).to(torch.int32).contiguous()
        mel_len = cum[:, -1].contiguous()
        L = int(mel_len.max())
        lib, stream = networks._runtime(feat)
        shape, blob = dec._shape(), dec._packed(lib, stream)
        outs = []
        for use_ws in (True, False):
            mel = torch.full((B, L, dec.n_mel_channels), float("nan"), dtype=torch.float32, device=device)
            ws, n = dec._workspace(lib, shape, B, L, feat.device) if use_ws else (None, 0)
            assert (n > 0) == (use_ws and cfg.dx2 == 256)
            lib.esmi_mel_decoder_f32(blob.data_ptr(), C.byref(shape), feat.data_ptr(), None, cum.data_ptr(), mel_len.data_ptr(), None, L, 1,
                                     B, T, L, mel.data_ptr(), ws.data_ptr() if ws is not None else None, n, stream)
            outs.append(mel.cpu())
        assert torch.isfinite(outs[0]).all()
        d = float((outs[0] - outs[1]).abs().max())
        assert d < 2e-6, (B, T, D, d)


def check_wrapper_and_scheduler(device):
    """model.py-shaped wrapper (`.phoneme2mel`, `.hifigan`, `model(x)`, `predict_step`, Lightning-dict load; model.py:155-164,
    demo.py:66-67) on a padded B > 1 batch and a B == 1 call, and the length-bucketed scheduler, against the oracle."""
    from oracle import oracle
    from efficientspeech_amd import EfficientSpeech, BucketedSynthesizer
    from efficientspeech_amd.synth import synth_phonemes
    cfg = CONFIGS["tiny"]
    sd = synth_state_dict(cfg, 1234)
    ckpt = {"state_dict": {"phoneme2mel." + k: torch.from_numpy(v) for k, v in sd.items()},
            "hyper_parameters": dict(depth=cfg.depth, n_blocks=cfg.n_blocks, block_depth=cfg.block_depth, reduction=cfg.reduction,
                                     head=cfg.head, embed_dim=cfg.embed_dim, kernel_size=cfg.kernel_size,
                                     decoder_kernel_size=cfg.decoder_kernel_size, expansion=cfg.expansion, lr=1e-3)}
    ckpt["state_dict"]["hifigan.conv_pre.bias"] = torch.zeros(4)           # vocoder keys present, no vocoder plugged in
    model = EfficientSpeech.load_from_checkpoint(ckpt).to(device)
    assert not model.training and model.hifigan is None and hasattr(model, "phoneme2mel")
    w = oracle.Weights(sd)
    rng = np.random.default_rng(8)
    # padded B > 1 batch, durations forced (random-init duration heads round to ~0-4 frames: keep the case non-degenerate)
    ids, mask = synth_phonemes(3, 23, 4, [23, 15, 6])
    dur = rng.integers(1, 5, size=(3, 23)).astype(np.int32)
    x = {"phoneme": torch.from_numpy(ids).to(device), "phoneme_mask": torch.from_numpy(mask).to(device),
         "duration_forced": torch.from_numpy(dur).to(device)}
    with torch.no_grad():
        enc = model.phoneme2mel.encoder._encode(x)
        wav, mel_len, duration = model(x)                                  # eval: predict_step; no vocoder -> (B, 80, L) mel
    o = oracle.phoneme2mel(cfg, w, ids, mask, pitch=enc["pitch"][..., 0].cpu().numpy(), energy=enc["energy"][..., 0].cpu().numpy(),
                           duration=dur)
    assert wav.shape == (3, 80, o.mel.shape[1]) and np.array_equal(mel_len.cpu().numpy(), o.mel_len)
    assert np.abs(wav.transpose(1, 2).cpu().numpy() - o.mel).max() < MEL_TOL
    np.testing.assert_allclose(duration.cpu().numpy(), o.duration, atol=PRED_TOL, rtol=0)
    # a vocoder plugged in: called on the channels-first mel, its output squeezed (model.py:161-162)
    seen = {}

    class FakeVocoder(torch.nn.Module):
        def forward(self, m):
            seen["shape"] = tuple(m.shape)
            return m.sum(1, keepdim=True)
    model.hifigan = FakeVocoder()
    with torch.no_grad():
        wav2, _, _ = model.predict_step(x)
    assert seen["shape"] == tuple(wav.shape) and wav2.shape == (3, wav.shape[2])
    model.hifigan = None
    # B == 1 (demo.py:66-67): no mask key at all
    ids1, _ = synth_phonemes(1, 19, 6)
    x1 = {"phoneme": torch.from_numpy(ids1).to(device), "duration_forced": torch.from_numpy(dur[:1, :19]).to(device)}
    with torch.no_grad():
        e1 = model.phoneme2mel.encoder._encode(x1)
        m1, l1, _ = model(x1)
    o1 = oracle.phoneme2mel(cfg, w, ids1, None, pitch=e1["pitch"][..., 0].cpu().numpy(), energy=e1["energy"][..., 0].cpu().numpy(),
                            duration=dur[:1, :19])
    assert np.array_equal(l1.cpu().numpy(), o1.mel_len) and np.abs(m1.transpose(1, 2).cpu().numpy() - o1.mel).max() < MEL_TOL
    # training mode dispatches to the teacher-forced dict (model.py:156)
    model.train()
    xt = dict(x)
    xt.update(pitch=torch.zeros((3, 23), device=device), energy=torch.zeros((3, 23), device=device),
              duration=torch.from_numpy(dur).to(device), mel_len=torch.from_numpy(dur.sum(1).astype(np.int32)).to(device))
    del xt["duration_forced"]
    with torch.no_grad():
        out = model(xt)
    assert isinstance(out, dict) and {"mel", "pitch", "energy", "duration", "mel_len", "features", "masks"} <= set(out)
    model.eval()
    # length-bucketed scheduler: every request answered once, in order, each batch = one reference-shaped forward
    lens = [9, 9, 5, 9, 5, 17, 16, 3, 9]
    seqs = [rng.integers(1, 150, size=n).astype(np.int32) for n in lens]
    sched = BucketedSynthesizer(model.phoneme2mel, max_batch=3, granularity=4)
    plan = sched.plan(lens)
    assert sorted(i for idx, _ in plan for i in idx) == list(range(len(lens))) and all(len(idx) <= 3 for idx, _ in plan)
    assert all(max(lens[i] for i in idx) == T and (T + 3) // 4 == (min(lens[i] for i in idx) + 3) // 4 for idx, T in plan)
    forced = {i: rng.integers(1, 4, size=n).astype(np.int32) for i, n in enumerate(lens)}

    def extra(idx, T):
        d = np.zeros((len(idx), T), np.int32)
        for r, i in enumerate(idx):
            d[r, :lens[i]] = forced[i]
        return {"duration_forced": torch.from_numpy(d).to(device)}
    res = sched(seqs, extra=extra)
    assert len(res) == len(lens)
    for idx, T in plan:                                # oracle on the same batches
        ids_b = np.zeros((len(idx), T), np.int32)
        d_b = np.zeros((len(idx), T), np.int32)
        for r, i in enumerate(idx):
            ids_b[r, :lens[i]] = seqs[i]
            d_b[r, :lens[i]] = forced[i]
        m_b = np.arange(T)[None, :] >= np.array([lens[i] for i in idx])[:, None]
        xb = {"phoneme": torch.from_numpy(ids_b).to(device), "duration_forced": torch.from_numpy(d_b).to(device)}
        if len(idx) > 1:
            xb["phoneme_mask"] = torch.from_numpy(m_b).to(device)
        with torch.no_grad():
            eb = model.phoneme2mel.encoder._encode(xb)
        ob = oracle.phoneme2mel(cfg, w, ids_b, m_b if len(idx) > 1 else None, pitch=eb["pitch"][..., 0].cpu().numpy(),
                                energy=eb["energy"][..., 0].cpu().numpy(), duration=d_b)
        for r, i in enumerate(idx):
            mel_i, dur_i = res[i]
            n = int(ob.mel_len[r])
            assert mel_i.shape == (n, 80) and dur_i.shape == (lens[i],)
            if n:
                assert np.abs(mel_i.cpu().numpy() - ob.mel[r, :n]).max() < MEL_TOL


def check_hifigan_golden(path, device):
    """One reference-generated HiFi-GAN fixture (tools/gen_golden_hifigan.py) through the module mirror / C-ABI."""
    from efficientspeech_amd.hifigan import HIFIGAN_CONFIGS, Generator, synth_hifigan_state_dict
    g = np.load(path)
    h = HIFIGAN_CONFIGS[str(g["config"])]
    voc = Generator(h)
    voc.load_state_dict({k: torch.from_numpy(v) for k, v in synth_hifigan_state_dict(h, 1234).items()}, strict=True)
    voc = voc.to(device).eval()
    mel = torch.from_numpy(g["mel"]).to(device)                    # (B, L, 80) channels-last, as the acoustic model emits it
    errs = []
    for fused in (True, False):                                    # one launch per ResBlock / one per convolution
        voc.fuse_resblocks = fused
        voc._cache.invalidate()
        with torch.no_grad():
            wav = voc(mel.transpose(1, 2))                         # the reference's calling convention: (B, 80, L)
        assert wav.shape == (mel.shape[0], 1, mel.shape[1] * h.hop)
        errs.append(float(np.abs(wav[:, 0].cpu().numpy() - g["wav"]).max()))
        assert errs[-1] < 5e-5, (fused, errs)
    return max(errs)
