"""Test infrastructure: the train=True forward and the loss of the reference (layers/networks.py:52-87, 151-165, 196-219, 233-244,
291-304, 336-434; model.py:167-216) written with PLAIN PyTorch ops on the parameters of our module mirror, so that torch.autograd
gives reference-class gradients at ANY size on the GPU box (where /root/reference does not exist).  It is itself pinned to the
reference-generated training fixtures (tests/test_train_step.py::test_torch_mirror_matches_reference_fixture); only tests use it."""
import torch
import torch.nn.functional as F

# Kink margins (SURVEY 7 "discrete decisions"): when a test sets TAPS = {} before a forward, every input of a non-smooth
# operation of the training graph -- the nine predictor ReLUs, the duration head's ReLU, |mel - target| of the L1 loss -- records
# min |value| here.  Two correct fp32 implementations may put an element closer to zero than their own rounding on opposite
# sides of the kink; its whole contribution to the gradients then moves (~1/rows).  A test case whose margins are below 1e-5
# is not a test of the kernels.
TAPS = None


def _tap(name, t, where=None):
    if TAPS is not None:
        v = t.detach().abs()
        if where is not None:
            v = v.masked_select(where)
        if v.numel():
            TAPS[name] = min(TAPS.get(name, float("inf")), float(v.min()))
    return t


def _conv(x, m):                         # channels-last wrapper around nn.Conv1d / ConvTranspose1d parameter containers
    if isinstance(m, torch.nn.Linear):
        return F.linear(x, m.weight, m.bias)
    if isinstance(m, torch.nn.ConvTranspose1d):
        return F.conv_transpose1d(x.transpose(1, 2), m.weight, m.bias, stride=m.stride[0], padding=m.padding[0]).transpose(1, 2)
    return F.conv1d(x.transpose(1, 2), m.weight, m.bias, stride=m.stride[0], padding=m.padding[0], groups=m.groups).transpose(1, 2)


def _ln(x, m):
    return F.layer_norm(x, (x.shape[-1],), m.weight, m.bias)


def _pool_mask(mask, n, n_out):          # blocks.py:51-57
    pool = int(torch.round(torch.tensor([n / n_out])).item())
    if pool <= 1:
        return mask
    pad = (-mask.shape[-1]) % pool
    if pad:
        mask = F.pad(mask, [0, pad], value=True)
    return mask.reshape(mask.shape[0], -1, pool).max(dim=-1).values[:, :n_out]


def _predictor(dec, fused, tag="predictor"):
    y = F.relu(_tap(tag + ".conv1", _conv(fused, dec.conv1[0])))
    y = F.relu(_tap(tag + ".norm1", _ln(y, dec.norm1)))
    y = F.relu(_tap(tag + ".conv2", _conv(y, dec.conv2[0])))
    pred = _conv(y, dec.linear)
    return (F.relu(_tap(tag + ".linear", pred)), _ln(y, dec.norm2)) if dec.duration else (pred, None)


def train_forward(net, x, stop_after_predictors=False, fused=None, preds=None):
    pe, dec = net.encoder, net.decoder
    if fused is not None:
        return _after_encoder(net, x, fused, preds["pitch"], preds["energy"], preds["duration"], preds["dur_feat"])
    phoneme = x["phoneme"].long()
    B, T = phoneme.shape
    mask = x["phoneme_mask"] if B > 1 else None
    h = F.embedding(phoneme, pe.encoder.embed.weight, padding_idx=0)
    feats = []
    for merge3, merge1, attn, ffn, norm1, norm2 in pe.encoder.attn_blocks:
        h = _conv(_conv(h, merge3), merge1)
        m = _pool_mask(mask, T, h.shape[1]) if mask is not None else None
        Bq, N, C = h.shape
        hd = attn.num_heads
        q, k, v = _conv(h, attn.qkv).reshape(Bq, N, 3, hd, C).permute(2, 0, 3, 1, 4).unbind(0)
        a = ((q @ k.transpose(-2, -1)) * (C // hd) ** -0.5).softmax(dim=-1)
        y = _conv((a @ v).transpose(1, 2).reshape(Bq, N, -1), attn.proj)
        h = _ln(y + h, norm1)
        if m is not None:
            h = h.masked_fill(m[..., None], 0)
        y = _conv(F.gelu(_conv(_conv(h, ffn.mlp1), ffn.conv)), ffn.mlp2)
        h = _ln(y + h, norm2)
        if m is not None:
            h = h.masked_fill(m[..., None], 0)
        feats.append(h)
    parts = []
    for f, (mlp, up) in zip(feats, pe.fuse.mlps):
        z = _conv(f, mlp)
        if isinstance(up, torch.nn.ConvTranspose1d):
            z = _conv(z, up)[:, :T]
        parts.append(z)
    fused = _conv(torch.cat(parts, -1), pe.fuse.fuse)
    if mask is not None:
        fused = fused.masked_fill(mask[..., None], 0)
    pitch_pred, _ = _predictor(pe.pitch_decoder, fused, "pitch")
    energy_pred, _ = _predictor(pe.energy_decoder, fused, "energy")
    dur_pred, dur_feat = _predictor(pe.duration_decoder, fused, "duration")
    if stop_after_predictors:
        return {"pitch": pitch_pred, "energy": energy_pred, "duration": dur_pred, "dur_feat": dur_feat, "fused": fused}
    return _after_encoder(net, x, fused, pitch_pred, energy_pred, dur_pred, dur_feat)


def _after_encoder(net, x, fused, pitch_pred, energy_pred, dur_pred, dur_feat):
    pe, dec = net.encoder, net.decoder
    B, T = x["phoneme"].shape
    mask = x["phoneme_mask"] if B > 1 else None
    pf = F.embedding(torch.bucketize(x["pitch"], pe.pitch_decoder.pitch_bins), pe.pitch_decoder.pitch_embedding.weight)
    ef = F.embedding(torch.bucketize(x["energy"], pe.energy_decoder.energy_bins), pe.energy_decoder.energy_embedding.weight)
    if mask is not None:
        pf, ef, dur_feat = (t.masked_fill(mask[..., None], 0) for t in (pf, ef, dur_feat))
    feat4 = torch.cat([fused, pf, ef, dur_feat], -1)
    dur = x["duration"].long()
    if mask is not None:
        dur = dur.masked_fill(mask, 0)
    L = int(x["mel"].shape[1])
    rows = []
    for f, d in zip(feat4, dur):
        r = f.repeat_interleave(d, dim=0)
        rows.append(F.pad(r, (0, 0, 0, L - r.shape[0])))
    features = torch.stack(rows)
    mel_len = dur.sum(1)
    skip = _ln(torch.tanh(_conv(features, dec.proj[0])), dec.proj[2])
    for convs, skip_norm in dec.blocks:
        z = skip
        for seq, norm in convs:
            z = _ln(torch.tanh(_conv(_conv(z, seq[0]), seq[1])), norm)
        skip = _ln(z + skip, skip_norm)
    mel = _conv(skip, dec.mel_linear)
    if mask is not None:
        mel = mel.masked_fill((torch.arange(L, device=mel.device)[None, :] >= mel_len[:, None])[..., None], 0)
    return {"mel": mel, "pitch": pitch_pred, "energy": energy_pred, "duration": dur_pred, "mel_len": mel_len}


def eval_forward(net, x):
    """Phoneme2Mel.forward(x, train=False) with the durations given (`duration_forced`, as the benchmark injects them): predicted
    pitch / energy are bucketised (networks.py:128-149 without targets), padding to the batch's longest utterance."""
    pe = net.encoder
    B, T = x["phoneme"].shape
    xx = dict(x)
    with torch.no_grad():
        # run the encoder side once to get the predictions, then reuse train_forward's data flow with them as "targets"
        dur = x["duration_forced"].long()
        if B > 1:
            dur = dur.masked_fill(x["phoneme_mask"], 0)
        L = int(dur.sum(1).max())
        xx.update(duration=dur, mel=torch.empty((B, L, 0), device=dur.device))
        probe = dict(xx, pitch=torch.zeros((B, T), device=dur.device), energy=torch.zeros((B, T), device=dur.device))
        first = train_forward(net, probe, stop_after_predictors=True)
        xx.update(pitch=first["pitch"].reshape(B, T), energy=first["energy"].reshape(B, T))
        out = train_forward(net, xx, fused=first["fused"], preds=first)
    return out["mel"], out["mel_len"], out["duration"]


def loss(out, x, y):
    mm, pm = ~x["mel_mask"][..., None], ~x["phoneme_mask"]
    sel = lambda t: t.reshape(pm.shape).masked_select(pm)      # noqa: E731
    _tap("l1", out["mel"] - y["mel"], mm.expand_as(out["mel"]))
    parts = [F.l1_loss(out["mel"].masked_select(mm), y["mel"].masked_select(mm)),
             F.mse_loss(sel(out["pitch"]), x["pitch"].masked_select(pm)), F.mse_loss(sel(out["energy"]), x["energy"].masked_select(pm)),
             F.mse_loss(torch.log(sel(out["duration"]) + 1), torch.log(x["duration"].masked_select(pm).to(out["duration"].dtype) + 1))]
    return parts, 10.0 * parts[0] + 2.0 * parts[1] + 2.0 * parts[2] + parts[3]
