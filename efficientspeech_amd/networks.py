"""Host-side mirror of the reference acoustic model API (layers/networks.py, layers/blocks.py).

Same class names, constructor arguments, `forward` signatures, return types and -- because the
sub-module nesting is reproduced -- the same `state_dict()` keys, so a Lightning
checkpoint's `phoneme2mel.*` tensors load with strict=True and `demo.py`-style callers work
unchanged.  None of the arithmetic happens here: every `forward` is a sequence of calls into the
esmi C-ABI (include/esmi.h -> libesmi.so, hand-written HIP for gfx950).  The torch layers below
are used purely as parameter containers; PyTorch supplies device memory and streams.

There is no CPU fallback: tensors must live on the MI355X and libesmi.so must be built.
"""
import contextlib
import ctypes as C
import os

import torch
from torch import nn

from . import _lib
from .config import N_SYMBOLS


# --------------------------------------------------------------------------- plumbing
def _runtime(t):
    """(lib, stream handle) for the device `t` lives on.  The product binds libesmi.so (HIP, gfx950) and nothing else; tests
    replace this hook to drive the CPU wave-simulator build of the same sources (tests/simlib.py)."""
    lib = _lib.load()
    if not t.is_cuda:
        raise RuntimeError("efficientspeech_amd: tensors must be on the GPU (no CPU fallback); "
                           "call .to('cuda') on the module and its inputs")
    return lib, torch.cuda.current_stream(t.device).cuda_stream


class _on_device_of:
    """Make the device of `t` current for the duration of a forward: kernel launches and per-device function attributes go
    to the CURRENT device, so a model on cuda:1 must not be driven while cuda:0 is current."""

    def __init__(self, t):
        self.idx = t.device.index if t.is_cuda else None
        self.prev = None

    def __enter__(self):
        if self.idx is not None:
            cur = torch.cuda.current_device()
            if cur != self.idx:
                self.prev = cur
                torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False


def _check_split_range(lib, stream, tensors, what):
    """Pack-time operand-range guard of the split-f16 build (include/esmi.h): every weight that feeds a split contraction
    must satisfy 2^8 |w| < 65504.  One device reduction per tensor, ONE host sync per checkpoint load."""
    limit = float(lib.esmi_split_weight_limit())
    tensors = [t for t in tensors if t is not None and t.numel() > 0]
    if limit == float("inf") or not tensors:
        return
    out = torch.zeros(len(tensors), dtype=torch.float32, device=tensors[0].device)
    for i, t in enumerate(tensors):
        t = _f32(t)
        lib.esmi_absmax_f32(_ptr(t), t.numel(), out[i:i + 1].data_ptr(), stream)
    m = float(out.max().item())
    if not m < limit:
        raise ValueError(f"{what}: max |weight| = {m:g} is outside the operand range of the split-f16 contractions "
                         f"(|w| < {limit:g}, include/esmi.h); use the exact-fp32 MFMA build for this checkpoint: "
                         f"ESMI_LIB=efficientspeech_amd/libesmi_fp32mfma.so")


def _ptr(t):
    return None if t is None else t.data_ptr()


def _f32(t):
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _mask_u8(mask):
    """bool (B,T) padding mask -> contiguous uint8 (utils/tools.py:43-51 convention: True = pad)."""
    if mask is None:
        return None
    return mask.detach().to(torch.bool).contiguous().view(torch.uint8)


def get_mask_from_lengths(lengths, max_len=None):
    """utils/tools.py:43-51: mask[b, t] = t >= lengths[b] (True marks padding)."""
    if max_len is None:
        max_len = int(torch.max(lengths).item())
    ids = torch.arange(0, max_len, device=lengths.device).unsqueeze(0)
    return ids >= lengths.unsqueeze(1)


class _PackCache:
    """Re-packs weights (tap-major convs, MFMA blob) only when a parameter changed."""

    def __init__(self):
        self.key = None
        self.val = None
        self.plist = None

    def get(self, params, build):
        """`params`: callable returning the parameters the packed form depends on.  The list is collected once (walking the
        module tree costs more host time than a whole forward step's launches: 118 `parameters()` walks = 0.1 ms per step
        before this was cached); a change of storage (`.to()`, `.float()`) or content (`load_state_dict`, in-place
        updates) of those Parameter objects is seen through (data_ptr, _version).  Replacing a Parameter OBJECT needs
        `invalidate()` (the module mirrors call it from `_apply` and after `load_state_dict`)."""
        if self.plist is None:
            self.plist = list(params())
        key = (_lib.generation(),) + tuple((p.data_ptr(), p._version) for p in self.plist)
        if key != self.key:
            self.val = build()
            self.key = key
        return self.val

    def invalidate(self):
        self.key = self.val = self.plist = None


class _PackedModule(nn.Module):
    """nn.Module whose `_PackCache` attributes are dropped whenever its parameters may have been replaced: `_apply`
    (`.to()`, `.cuda()`, `.float()` ...) and `load_state_dict` (incl. `assign=True`)."""

    def __init__(self):
        super().__init__()
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module._drop_packed())

    def _drop_packed(self):
        for v in self.__dict__.values():
            if isinstance(v, _PackCache):
                v.invalidate()

    def _apply(self, fn, *args, **kwargs):
        self._drop_packed()
        return super()._apply(fn, *args, **kwargs)


def _pack_conv(lib, stream, w, transposed=False):
    """(Cout,Cin,k) [or ConvTranspose (Cin,Cout,k)] -> tap-major (k,Cout,Cin) on device."""
    w = _f32(w)
    a, b, k = w.shape
    cout, cin = (b, a) if transposed else (a, b)
    dst = torch.empty((k, cout, cin), dtype=torch.float32, device=w.device)
    if transposed:
        lib.esmi_pack_convT_weight_f32(_ptr(w), _ptr(dst), cin, cout, k, stream)
    else:
        lib.esmi_pack_conv_weight_f32(_ptr(w), _ptr(dst), cout, cin, k, stream)
    return dst, w


def _pack_bfrag(lib, stream, w):
    """(n,k) Linear weight or tap-major (taps,n,k) conv weight -> MFMA B-fragment order (include/esmi.h)."""
    w = _f32(w)
    taps, n, k = (1, *w.shape) if w.dim() == 2 else w.shape
    dst = torch.empty(lib.esmi_pack_bfrag_floats(n, k, taps), dtype=torch.float32, device=w.device)
    lib.esmi_pack_bfrag_f32(_ptr(w), _ptr(dst), n, k, taps, stream)
    return dst


# --------------------------------------------------------------------------- encoder sub-modules
class SelfAttention(nn.Module):
    """blocks.py:32-71 (qkv bias-free; every head is full width: qkv = 3*h*dim).  Inside `Encoder` the block runs fused
    (esmi_encoder_block_f32); `forward` is the module-level API through esmi_self_attention_f32."""

    def __init__(self, dim, num_heads=1, qkv_bias=False):
        super().__init__()
        assert dim % num_heads == 0, 'dim should be divisible by num_heads'
        if qkv_bias:
            raise NotImplementedError("qkv_bias=True: the reference never sets it (networks.py:43) and the kernels take a bias-free qkv")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3 * num_heads, bias=qkv_bias)
        self.proj = nn.Linear(dim * num_heads, dim)

    def forward(self, x, mask=None, pool=1):
        """x (B,N,C); mask bool (B,T) or None -> (proj(softmax(q k^T scale) v) (B,N,C), attn_mask bool (B,N,C) or None).
        As in the reference the scores are NOT masked; the returned mask is the max-pooled padding mask (blocks.py:51-69)."""
        x = _f32(x)
        with _on_device_of(x):
            lib, stream = _runtime(x)
            B, N, Cc = x.shape
            out = torch.empty_like(x)
            ws_bytes = lib.esmi_self_attention_workspace_bytes(B, N, Cc, self.num_heads)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
            lib.esmi_self_attention_f32(_ptr(_f32(self.qkv.weight)), _ptr(_f32(self.proj.weight)), _ptr(_f32(self.proj.bias)),
                                        B, N, Cc, self.num_heads, _ptr(x), _ptr(out), _ptr(ws), ws_bytes, stream)
            attn_mask = None
            if mask is not None:
                m8 = _mask_u8(mask)
                if pool > 1:
                    T = m8.shape[-1]
                    n_out = (T + pool - 1) // pool
                    pm = torch.empty((B, n_out), dtype=torch.uint8, device=x.device)
                    lib.esmi_pool_mask_u8(_ptr(m8), B, T, int(pool), _ptr(pm), n_out, stream)
                    m8 = pm
                attn_mask = m8.bool().unsqueeze(-1).expand(-1, -1, Cc)
            return out, attn_mask


class MixFFN(nn.Module):
    """blocks.py:8-29: Linear -> dense Conv1d k3 -> GELU (exact erf) -> Linear."""

    def __init__(self, dim, expansion_factor):
        super().__init__()
        hidden = dim * expansion_factor
        self.expansion_factor = expansion_factor
        self.mlp1 = nn.Linear(dim, hidden)
        self.conv = nn.Conv1d(hidden, hidden, 3, padding=1)
        self.mlp2 = nn.Linear(hidden, dim)

    def forward(self, x):
        """x (B,N,C) -> (B,N,C) through esmi_mixffn_f32."""
        x = _f32(x)
        with _on_device_of(x):
            lib, stream = _runtime(x)
            B, N, Cc = x.shape
            cw, _ = _pack_conv(lib, stream, self.conv.weight)
            out = torch.empty_like(x)
            ws_bytes = lib.esmi_mixffn_workspace_bytes(B, N, Cc, self.expansion_factor)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
            lib.esmi_mixffn_f32(_ptr(_f32(self.mlp1.weight)), _ptr(_f32(self.mlp1.bias)), _ptr(cw), _ptr(_f32(self.conv.bias)),
                                _ptr(_f32(self.mlp2.weight)), _ptr(_f32(self.mlp2.bias)), B, N, Cc, self.expansion_factor,
                                _ptr(x), _ptr(out), _ptr(ws), ws_bytes, stream)
            return out


class Encoder(_PackedModule):
    """Phoneme encoder pyramid (networks.py:15-87): per block merge convs -> attention -> MixFFN."""

    def __init__(self, depth=2, embed_dim=128, kernel_size=3, expansion=1, reduction=4, head=1):
        super().__init__()
        dim = embed_dim // reduction
        self.depth, self.embed_dim, self.expansion = depth, embed_dim, expansion
        self.dim_outs = [dim * 2 ** i for i in range(depth)]
        self.dim_ins = [embed_dim] + self.dim_outs[:-1]
        self.heads = [head * (i + 1) for i in range(depth)]
        self.kernels = [kernel_size - (2 if i > 0 else 0) for i in range(depth)]
        self.strides = [1] + [2] * (depth - 1)
        self.embed = nn.Embedding(N_SYMBOLS + 1, embed_dim, padding_idx=0)
        self.attn_blocks = nn.ModuleList([
            nn.ModuleList([
                nn.Conv1d(ci, ci, kernel_size=k, stride=s, padding=k // 2, bias=False),   # dense, NOT depthwise
                nn.Conv1d(ci, co, kernel_size=1, bias=False),
                SelfAttention(co, num_heads=h),
                MixFFN(co, expansion),
                nn.LayerNorm(co),
                nn.LayerNorm(co),
            ]) for ci, co, h, k, s in zip(self.dim_ins, self.dim_outs, self.heads, self.kernels, self.strides)])
        self._cache = _PackCache()

    def get_feature_dims(self):
        return self.dim_outs

    def block_len(self, T, i):
        n = T
        for j in range(i + 1):
            k = self.kernels[j]
            n = (n + 2 * (k // 2) - k) // self.strides[j] + 1
        return n

    def _packed(self, lib, stream):
        def build():
            out = []
            for merge, merge1, attn, ffn, n1, n2 in self.attn_blocks:
                keep = []
                mw, _ = _pack_conv(lib, stream, merge.weight)
                cw, _ = _pack_conv(lib, stream, ffn.conv.weight)
                t = dict(merge_w=mw, merge1_w=_f32(merge1.weight), qkv_w=_f32(attn.qkv.weight),
                         proj_w=_f32(attn.proj.weight), proj_b=_f32(attn.proj.bias),
                         mlp1_w=_f32(ffn.mlp1.weight), mlp1_b=_f32(ffn.mlp1.bias), conv_w=cw,
                         conv_b=_f32(ffn.conv.bias), mlp2_w=_f32(ffn.mlp2.weight), mlp2_b=_f32(ffn.mlp2.bias),
                         ln1_g=_f32(n1.weight), ln1_b=_f32(n1.bias), ln2_g=_f32(n2.weight), ln2_b=_f32(n2.bias))
                # weight-folded attention for the one-kernel-per-op plan (include/esmi.h): M_h = Wq_h^T Wk_h, O_h = Wv_h^T Wp_h^T, in fp64
                hh, cc = attn.num_heads, t["proj_w"].shape[0]
                wqkv = t["qkv_w"].double().view(3, hh, cc, cc)                       # [s][h][out][in]
                wp = t["proj_w"].double().view(cc, hh, cc)                            # [out][h][in]
                t["qk_w"] = torch.einsum("hoj,hoi->hji", wqkv[1], wqkv[0]).reshape(hh * cc, cc).float().contiguous()   # rows [h][j]: sum_o Wk[o][j] Wq[o][i]
                t["vo_w"] = torch.einsum("ohm,hmi->ohi", wp, wqkv[2]).reshape(cc, hh * cc).float().contiguous()        # [o][h][i]: sum_m Wp[o][h][m] Wv[m][i]
                # MixFFN's Linear folded into its k = 3 conv (include/esmi.h, ffn_cw), in fp64: W'[j] = conv[j] @ mlp1; the Linear's bias
                # reaches a position through every tap that lies inside the sequence
                if os.environ.get("ESMI_FOLD_FFN", "1") != "0":                  # (the environment switch: development A/B)
                    cw64 = cw.double()                                            # (3, eC, eC) tap-major
                    e = torch.einsum("jom,m->jo", cw64, t["mlp1_b"].double())    # what tap j carries of the Linear's bias
                    t["ffn_cw"] = torch.einsum("jom,mi->joi", cw64, t["mlp1_w"].double()).float().contiguous()
                    t["ffn_cb"] = (t["conv_b"].double() + e.sum(0)).float().contiguous()
                    t["ffn_cb_first"], t["ffn_cb_last"] = e[0].float().contiguous(), e[2].float().contiguous()
                for name in ("qkv_w", "proj_w", "mlp1_w", "conv_w", "mlp2_w", "qk_w", "vo_w") + (("ffn_cw",) if "ffn_cw" in t else ()):
                    t[name + "p"] = _pack_bfrag(lib, stream, t[name])

                k, cin, co = mw.shape[0], mw.shape[2], merge1.weight.shape[0]
                comp = torch.empty((k, co, cin), dtype=torch.float32, device=mw.device)   # merge1 o merge as one conv
                lib.esmi_compose_merge_f32(_ptr(mw), _ptr(t["merge1_w"]), k, cin, co, _ptr(comp), stream)
                t["merge_cwp"] = _pack_bfrag(lib, stream, comp)
                if len(out) == 0 and os.environ.get("ESMI_FOLD_EMBED", "1") != "0":   # (the environment switch: development A/B)
                    # block 0: embedding, merge conv and merge 1x1 are linear with nothing in between -> one table per tap,
                    # E_j = embed @ (merge1 @ merge[j])^T, in fp64 (esmi.h, esmi_encoder_block_weights.emb_conv)
                    t["emb_conv"] = torch.einsum("vi,jmi,om->jvo", self.embed.weight.detach().double(), mw.double(),
                                                 t["merge1_w"].double().reshape(co, cin)).float().contiguous()
                _check_split_range(lib, stream, [comp, t["qkv_w"], t["proj_w"], t["mlp1_w"], cw, t["mlp2_w"], t["qk_w"], t["vo_w"]]
                                   + ([t["ffn_cw"]] if "ffn_cw" in t else []),
                                   "encoder block weights")
                # measured (same box, per-op plan): base ES (2 and 4 heads) 9.01 -> 8.48 ms/step; small ES block 0 (ONE head: the
                # projections shrink 192 -> 64 columns only) 2.22 -> 2.25: folded only where there are heads to share the keys / values
                # (the per-op plan uses them for blocks with two heads or more only -- esmi_encoder_block_f32 decides; the whole-block chain
                #  kernels for every block)
                if os.environ.get("ESMI_FOLD_ATTN", "1") == "0":             # (the environment switch: development A/B)
                    for name in ("qk_w", "qk_wp", "vo_w", "vo_wp"):
                        del t[name]
                keep.extend(t.values())
                out.append((_lib.EncoderBlockWeights(**{k: _ptr(v) for k, v in t.items()}), keep))
            return out, _f32(self.embed.weight)
        return self._cache.get(lambda: list(self.parameters()), build)

    def forward(self, phoneme, mask=None):
        with _on_device_of(self.embed.weight):
            return self._forward(phoneme, mask)

    def _forward(self, phoneme, mask=None):
        """phoneme int (B,T); mask bool (B,T) or None -> ([f_0..f_{depth-1}], decoder_mask (B,T,dim) or None)."""
        feats, masks = self._run(phoneme, mask)
        decoder_mask = None
        if mask is not None:
            decoder_mask = masks[0].bool().unsqueeze(-1).expand(-1, -1, self.dim_outs[0])
        return feats, decoder_mask

    def _run(self, phoneme, mask):
        lib, stream = _runtime(self.embed.weight)
        (blocks, embed) = self._packed(lib, stream)
        dev = embed.device
        ids = phoneme.detach().to(device=dev, dtype=torch.int32).contiguous()
        B, T = ids.shape
        m8 = _mask_u8(mask)
        feats, masks = [], []
        x_in, n_in = None, T
        for i, (wts, _keep) in enumerate(blocks):
            n = self.block_len(T, i)
            pool = 1
            if m8 is not None:                                   # networks.py:69-70 + blocks.py:51-57
                pool = int(round(T / n))                         # torch.round == Python round: half to even
                if (T + pool - 1) // pool != n:
                    raise RuntimeError(f"pooled mask length {(T + pool - 1) // pool} != sequence length {n}")
            # the padding mask is pooled on the fly inside the block (mask_pool): no pooled copy is materialised
            shape = _lib.EncoderBlockShape(B, n_in, self.dim_ins[i], self.dim_outs[i], self.heads[i], self.kernels[i],
                                           self.strides[i], self.expansion, N_SYMBOLS + 1, pool, T, _lib.current_plan())
            ws_bytes = lib.esmi_encoder_block_workspace_bytes(C.byref(shape))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            x_out = torch.empty((B, n, self.dim_outs[i]), dtype=torch.float32, device=dev)
            lib.esmi_encoder_block_f32(C.byref(wts), C.byref(shape), _ptr(ids) if i == 0 else None,
                                       _ptr(embed) if i == 0 else None, _ptr(x_in), _ptr(m8), _ptr(x_out),
                                       _ptr(ws), ws_bytes, stream)
            feats.append(x_out)
            masks.append(m8 if pool == 1 else None)
            x_in, n_in = x_out, n
        return feats, masks


class AcousticDecoder(nn.Module):
    """Pitch / energy / duration predictor parameters (networks.py:90-125)."""

    def __init__(self, dim, pitch_stats=None, energy_stats=None, n_mel_channels=80, duration=False):
        super().__init__()
        self.n_mel_channels = n_mel_channels
        self.duration = duration
        self.conv1 = nn.Sequential(nn.Conv1d(dim, dim, kernel_size=3, padding=1), nn.ReLU())
        self.norm1 = nn.LayerNorm(dim)
        self.conv2 = nn.Sequential(nn.Conv1d(dim, dim, kernel_size=3, padding=1), nn.ReLU())
        self.norm2 = nn.LayerNorm(dim)
        self.linear = nn.Linear(dim, 1)
        self.pitch_bins = self.pitch_embedding = self.energy_bins = self.energy_embedding = None
        if pitch_stats is not None:
            lo, hi = pitch_stats
            self.pitch_bins = nn.Parameter(torch.linspace(lo, hi, dim - 1), requires_grad=False)
            self.pitch_embedding = nn.Embedding(dim, dim)
        if energy_stats is not None:
            lo, hi = energy_stats
            self.energy_bins = nn.Parameter(torch.linspace(lo, hi, dim - 1), requires_grad=False)
            self.energy_embedding = nn.Embedding(dim, dim)

    def _weights(self, lib, stream):
        c1, _ = _pack_conv(lib, stream, self.conv1[0].weight)
        c2, _ = _pack_conv(lib, stream, self.conv2[0].weight)
        bins = self.pitch_bins if self.pitch_bins is not None else self.energy_bins
        emb = self.pitch_embedding if self.pitch_embedding is not None else self.energy_embedding
        t = dict(conv1_w=c1, conv1_b=_f32(self.conv1[0].bias), ln1_g=_f32(self.norm1.weight), ln1_b=_f32(self.norm1.bias),
                 conv2_w=c2, conv2_b=_f32(self.conv2[0].bias), ln2_g=_f32(self.norm2.weight), ln2_b=_f32(self.norm2.bias),
                 lin_w=_f32(self.linear.weight), lin_b=_f32(self.linear.bias),
                 bins=None if bins is None else _f32(bins), emb=None if emb is None else _f32(emb.weight))
        t["conv1_wp"], t["conv2_wp"] = _pack_bfrag(lib, stream, c1), _pack_bfrag(lib, stream, c2)
        _check_split_range(lib, stream, [c1, c2], "variance predictor weights")
        return _lib.PredictorWeights(**{k: _ptr(v) for k, v in t.items()}), list(t.values())

    def forward(self, fused_features):
        """networks.py:151-165 through esmi_acoustic_decoder_f32: (B,T,dim) -> y (B,T,1) [duration: (relu(y), features = LN2(.))];
        `linear` reads the PRE-norm2 tensor.  (PhonemeEncoder runs all three predictors in one fused kernel instead.)"""
        x = _f32(fused_features)
        with _on_device_of(x):
            lib, stream = _runtime(x)
            B, T, dim = x.shape
            w, _keep = self._weights(lib, stream)
            pred = torch.empty((B, T, 1), dtype=torch.float32, device=x.device)
            feats = torch.empty((B, T, dim), dtype=torch.float32, device=x.device) if self.duration else None
            ws_bytes = lib.esmi_variance_adaptor_workspace_bytes(B, T, dim)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
            lib.esmi_acoustic_decoder_f32(C.byref(w), dim, B, T, int(self.duration), _ptr(x), dim, _ptr(pred), _ptr(feats),
                                          _ptr(ws), ws_bytes, stream)
            return (pred, feats) if self.duration else pred

    def get_embedding(self, pred, target, mask, control=1.):
        """networks.py:128-149: embedding(bucketize(target if target is not None else pred, bins)); shape = v.shape + (dim,)."""
        bins = self.pitch_bins if self.pitch_bins is not None else self.energy_bins
        emb = self.pitch_embedding if self.pitch_embedding is not None else self.energy_embedding
        if emb is None:
            return None
        v = _f32(target if target is not None else pred)
        with _on_device_of(v):
            lib, stream = _runtime(v)
            dim = emb.weight.shape[1]
            out = torch.empty(tuple(v.shape) + (dim,), dtype=torch.float32, device=v.device)
            if v.numel():
                lib.esmi_bucket_embedding_f32(_ptr(v), _ptr(_f32(bins)), _ptr(_f32(emb.weight)), v.numel(), dim, _ptr(out), None, stream)
            return out

    get_pitch_embedding = get_energy_embedding = get_embedding


class Fuse(_PackedModule):
    """Fuses the pyramid features back to phoneme rate (networks.py:168-219)."""

    def __init__(self, dims, kernel_size=3):
        super().__init__()
        assert len(dims) > 0
        dim = dims[0]
        self.dims, self.kernel_size = list(dims), kernel_size
        self.mlps = nn.ModuleList([
            nn.ModuleList([nn.Linear(d, dim),
                           nn.ConvTranspose1d(dim, dim, kernel_size=kernel_size, stride=d // dim)
                           if d // dim > 1 else nn.Identity()]) for d in dims])
        self.fuse = nn.Linear(dim * len(dims), dim)
        self._cache = _PackCache()

    def _packed(self, lib, stream):
        def build():
            w = _lib.FuseWeights()
            keep = []
            for i, (lin, up) in enumerate(self.mlps):
                lw, lb = _f32(lin.weight), _f32(lin.bias)
                lwp = _pack_bfrag(lib, stream, lw)
                keep += [lw, lb, lwp]
                w.mlp_w[i], w.mlp_b[i], w.mlp_wp[i] = _ptr(lw), _ptr(lb), _ptr(lwp)
                if isinstance(up, nn.ConvTranspose1d):
                    uw, _ = _pack_conv(lib, stream, up.weight, transposed=True)
                    ub = _f32(up.bias)
                    uwp = _pack_bfrag(lib, stream, uw)
                    keep += [uw, ub, uwp]
                    w.up_w[i], w.up_b[i], w.up_wp[i] = _ptr(uw), _ptr(ub), _ptr(uwp)
            fw, fb = _f32(self.fuse.weight), _f32(self.fuse.bias)
            fwp = _pack_bfrag(lib, stream, fw)
            keep += [fw, fb, fwp]
            w.fuse_w, w.fuse_b, w.fuse_wp = _ptr(fw), _ptr(fb), _ptr(fwp)
            _check_split_range(lib, stream, [t for t in keep if t.dim() >= 2 and t.dtype == torch.float32], "Fuse weights")
            return w, keep
        return self._cache.get(lambda: list(self.parameters()), build)

    def _run(self, feats, m8, out=None, ld_out=None):
        lib, stream = _runtime(self.fuse.weight)
        w, _keep = self._packed(lib, stream)
        dim, depth = self.dims[0], len(self.dims)
        B, T = feats[0].shape[0], feats[0].shape[1]
        dev = feats[0].device
        if out is None:
            out = torch.empty((B, T, dim), dtype=torch.float32, device=dev)
            ld_out = dim
        ws_bytes = lib.esmi_fuse_workspace_bytes(B, T, dim, depth)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        fp = (C.c_void_p * depth)(*[_ptr(f) for f in feats])
        ni = (C.c_int * depth)(*[f.shape[1] for f in feats])
        lib.esmi_fuse_f32(C.byref(w), depth, dim, self.kernel_size, B, T, fp, ni, _ptr(m8), _ptr(out), ld_out,
                          _ptr(ws), ws_bytes, stream)
        return out

    def forward(self, features, mask=None):
        with _on_device_of(self.fuse.weight):
            return self._forward(features, mask)

    def _forward(self, features, mask=None):
        """features: list of (B,N_i,dim*2^i); mask: (B,T,dim) bool or None -> (B,T,dim)."""
        m8 = None if mask is None else _mask_u8(mask[..., 0])
        return self._run([_f32(f) for f in features], m8)


class FeatureUpsampler(nn.Module):
    """Length regulator (networks.py:222-258) as a device-side scan + gather (no per-utterance host sync)."""

    def forward(self, fused_features, fused_masks, duration, max_mel_len=None):
        with _on_device_of(fused_features):
            return self._forward(fused_features, fused_masks, duration, max_mel_len)

    def _forward(self, fused_features, fused_masks, duration, max_mel_len=None):
        """fused_features (B,T,C); fused_masks (B,T,C) bool; duration (B,T[,1]) -> features, masks, mel_len."""
        feat = _f32(fused_features)
        lib, stream = _runtime(feat)
        B, T, Cc = feat.shape
        dev = feat.device
        dur = duration.detach().reshape(B, T).to(device=dev, dtype=torch.int32).contiguous()
        cum = torch.empty((B, T), dtype=torch.int32, device=dev)
        mel_len = torch.empty((B,), dtype=torch.int32, device=dev)
        lmax = torch.empty((1,), dtype=torch.int32, device=dev)
        lib.esmi_length_regulate_i32(_ptr(dur), B, T, _ptr(cum), _ptr(mel_len), _ptr(lmax), stream)
        L = int(max_mel_len) if max_mel_len is not None else int(lmax.item())
        fm = None if fused_masks is None else _mask_u8(fused_masks[..., 0])
        features = torch.empty((B, L, Cc), dtype=torch.float32, device=dev)
        masks = torch.empty((B, L), dtype=torch.uint8, device=dev)
        if L > 0:
            lib.esmi_upsample_f32(_ptr(feat), _ptr(fm), _ptr(cum), B, T, Cc, L, _ptr(features), _ptr(masks), stream)
        return features, masks.bool().unsqueeze(-1).expand(-1, -1, Cc), mel_len


class MelDecoder(_PackedModule):
    """Mel spectrogram decoder (networks.py:261-304), one fused HIP kernel per call."""

    def __init__(self, dim, kernel_size=5, n_mel_channels=80, n_blocks=2, block_depth=2):
        super().__init__()
        self.n_mel_channels = n_mel_channels
        self.kernel_size, self.n_blocks, self.block_depth = kernel_size, n_blocks, block_depth
        self.dim_x4 = 4 * dim
        self.dim_x2 = dx2 = min(4 * dim, 256)
        self.proj = nn.Sequential(nn.Linear(self.dim_x4, dx2), nn.Tanh(), nn.LayerNorm(dx2))
        self.blocks = nn.ModuleList([
            nn.ModuleList([
                nn.ModuleList([
                    nn.ModuleList([
                        nn.Sequential(nn.Conv1d(dx2, dx2, groups=dx2, kernel_size=kernel_size, padding=kernel_size // 2),
                                      nn.Conv1d(dx2, dx2, kernel_size=1), nn.Tanh()),
                        nn.LayerNorm(dx2)]) for _ in range(block_depth)]),
                nn.LayerNorm(dx2)]) for _ in range(n_blocks)])
        self.mel_linear = nn.Linear(dx2, n_mel_channels)
        self._cache = _PackCache()
        self._head_cache = _PackCache()
        self.timing = None      # bench.py sets this to a list to collect (start, end) HIP events per timed launch
        self.timing_every = 1   # ... on every n-th launch only (an event pair costs ~10 us of pipeline drain per step)
        self._launches = 0

    def _shape(self):
        return _lib.DecoderShape(self.dim_x4, self.dim_x2, self.kernel_size, self.n_blocks, self.block_depth,
                                 self.n_mel_channels)

    def _packed(self, lib, stream):
        def build():
            shape = self._shape()
            nbytes = lib.esmi_mel_decoder_blob_bytes(C.byref(shape))
            if nbytes == 0:
                raise RuntimeError(f"mel decoder shape not supported by the HIP kernel: d4={self.dim_x4} "
                                   f"dx2={self.dim_x2} k={self.kernel_size} ({self.n_blocks}x{self.block_depth})")
            w = _lib.DecoderWeights()
            keep = []

            def put(name, t, idx=None):
                t = _f32(t)
                keep.append(t)
                if idx is None:
                    setattr(w, name, _ptr(t))
                else:
                    getattr(w, name)[idx] = _ptr(t)
            put("proj_w", self.proj[0].weight); put("proj_b", self.proj[0].bias)
            put("proj_ln_g", self.proj[2].weight); put("proj_ln_b", self.proj[2].bias)
            layer = 0
            for b, (convs, skip_norm) in enumerate(self.blocks):
                for conv, norm in convs:
                    put("dw_w", conv[0].weight, layer); put("dw_b", conv[0].bias, layer)
                    put("pw_w", conv[1].weight, layer); put("pw_b", conv[1].bias, layer)
                    put("ln_g", norm.weight, layer); put("ln_b", norm.bias, layer)
                    layer += 1
                put("skip_g", skip_norm.weight, b); put("skip_b", skip_norm.bias, b)
            put("mel_w", self.mel_linear.weight); put("mel_b", self.mel_linear.bias)
            blob = torch.empty(nbytes // 4, dtype=torch.float32, device=keep[0].device)
            _check_split_range(lib, stream, [t for t in keep if t.dim() >= 2 and t.shape[1] > 1], "mel decoder weights")
            lib.esmi_mel_decoder_pack_f32(C.byref(w), C.byref(shape), _ptr(blob), stream)
            return blob
        return self._cache.get(lambda: list(self.parameters()), build)

    def forward(self, features):
        with _on_device_of(self.mel_linear.weight):
            return self._forward(features)

    def _forward(self, features):
        """features (B,L,4*dim) -> mel (B,L,n_mel).  (Direct mode: rows exactly as given.)"""
        x = _f32(features)
        lib, stream = _runtime(x)
        B, L, _ = x.shape
        mel = torch.empty((B, L, self.n_mel_channels), dtype=torch.float32, device=x.device)
        if L > 0:
            shape = self._shape()
            ws, ws_bytes = self._workspace(lib, shape, B, L, x.device)
            lib.esmi_mel_decoder_f32(_ptr(self._packed(lib, stream)), C.byref(shape), _ptr(x), None, None, None, None, L, 0,
                                     B, 0, L, _ptr(mel), _ptr(ws), ws_bytes, stream)
        return mel

    @staticmethod
    def _workspace(lib, shape, B, L, dev):
        """Scratch for the dx2 = 256 kernel's carried rows (esmi_mel_decoder_workspace_bytes; nothing for dx2 = 128)."""
        n = lib.esmi_mel_decoder_workspace_bytes(C.byref(shape), B, L)
        return (torch.empty(n, dtype=torch.uint8, device=dev), n) if n else (None, 0)

    def _head(self, lib, stream):
        """The decoder's first stage (proj Linear + Tanh + LN) for the encoder side: it is row-wise, so it runs once per
        PHONEME there -- inside the fused variance-adaptor kernel (tiny ES: `proj_wp`) or as one GEMM launch behind it (every
        other size: `proj_w`) -- and the decoder only gathers.  None when the width is not one the GEMM's LayerNorm epilogue serves."""
        if self.dim_x2 not in (32, 64, 128, 256):
            return None
        if os.environ.get("ESMI_HEAD_GEMM", "1") == "0" and not (self.dim_x4 == 128 and self.dim_x2 == 128):
            return None                      # (development A/B: the decoder runs its first stage itself, at frame rate)

        def build():
            t = dict(proj_b=_f32(self.proj[0].bias), ln_g=_f32(self.proj[2].weight), ln_b=_f32(self.proj[2].bias),
                     proj_w=_f32(self.proj[0].weight))
            t["proj_wp"] = _pack_bfrag(lib, stream, self.proj[0].weight)     # (the chain kernel's operand order = the GEMM's pre-split form)
            return _lib.DecoderHead(d4=self.dim_x4, dx2=self.dim_x2, **{k: _ptr(v) for k, v in t.items()}), list(t.values())
        return self._head_cache.get(lambda: list(self.proj.parameters()), build)

    def _fused(self, feat, cum, mel_len, lmax_dev, lmax_host, apply_mask, L_out, h0=None):
        """Length-regulator gather fused into the decoder: feat (B,T,d4) phoneme-rate; h0 (B,T,dx2) = the first stage's
        output at phoneme rate when the encoder side computed it."""
        lib, stream = _runtime(feat)
        B, T, _ = feat.shape
        mel = torch.empty((B, L_out, self.n_mel_channels), dtype=torch.float32, device=feat.device)
        if L_out > 0:
            shape = self._shape()
            blob = self._packed(lib, stream)
            ev = None
            self._launches += 1
            if self.timing is not None and feat.is_cuda and self._launches % self.timing_every == 0:   # events on the launch stream
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            ws, ws_bytes = self._workspace(lib, shape, B, L_out, feat.device)
            lib.esmi_mel_decoder_f32(_ptr(blob), C.byref(shape), _ptr(feat), _ptr(h0), _ptr(cum), _ptr(mel_len), _ptr(lmax_dev),
               int(lmax_host), int(apply_mask), B, T, L_out, _ptr(mel), _ptr(ws), ws_bytes, stream)
            if ev is not None:
                ev[1].record()
                self.timing.append(ev)
        return mel


class PhonemeEncoder(_PackedModule):
    """Phonemes -> variance-adapted, length-regulated acoustic features (networks.py:307-401)."""

    def __init__(self, pitch_stats=None, energy_stats=None, depth=2, reduction=4, head=1, embed_dim=128,
                 kernel_size=3, expansion=1):
        super().__init__()
        self.encoder = Encoder(depth=depth, reduction=reduction, head=head, embed_dim=embed_dim,
                               kernel_size=kernel_size, expansion=expansion)
        dim = embed_dim // reduction
        self.dim = dim
        self.fuse = Fuse(self.encoder.get_feature_dims(), kernel_size=kernel_size)
        self.feature_upsampler = FeatureUpsampler()
        self.pitch_decoder = AcousticDecoder(dim, pitch_stats=pitch_stats)
        self.energy_decoder = AcousticDecoder(dim, energy_stats=energy_stats)
        self.duration_decoder = AcousticDecoder(dim, duration=True)
        self._cache = _PackCache()

    def _predictors(self, lib, stream):
        def build():
            return [d._weights(lib, stream) for d in (self.pitch_decoder, self.energy_decoder, self.duration_decoder)]
        return self._cache.get(lambda: [p for d in (self.pitch_decoder, self.energy_decoder, self.duration_decoder)
                                        for p in d.parameters()], build)

    def _encode(self, x, train=False, need_lmax=True, head=None):
        """Everything up to (and including) the duration scan; nothing frame-rate is materialised."""
        phoneme = x["phoneme"]
        B = phoneme.shape[0]
        phoneme_mask = x["phoneme_mask"] if B > 1 else None           # KeyError for B>1, as the reference (:338)
        dev = self.encoder.embed.weight.device
        lib, stream = _runtime(self.encoder.embed.weight)
        feats, bmasks = self.encoder._run(phoneme, phoneme_mask)
        T, dim = feats[0].shape[1], self.dim
        m8 = bmasks[0]
        feat = torch.empty((B, T, 4 * dim), dtype=torch.float32, device=dev)
        fw, _kf = self.fuse._packed(lib, stream)
        (pw, _k0), (ew, _k1), (dw, _k2) = self._predictors(lib, stream)

        def tgt(key, dtype):
            if not train:
                return None
            return x[key].detach().reshape(B, T).to(device=dev, dtype=dtype).contiguous()
        pitch_t, energy_t = tgt("pitch", torch.float32), tgt("energy", torch.float32)
        dur_t = tgt("duration", torch.int32)
        if not train and "duration_forced" in x:                       # extension: inject durations at inference
            dur_t = x["duration_forced"].detach().reshape(B, T).to(device=dev, dtype=torch.int32).contiguous()
        preds = torch.empty((3, B, T, 1), dtype=torch.float32, device=dev)
        idxs = torch.empty((2, B, T), dtype=torch.int32, device=dev)
        dur = torch.empty((B, T), dtype=torch.int32, device=dev)
        depth = len(feats)
        ws_bytes = lib.esmi_fuse_variance_adaptor_workspace_bytes(B, T, dim, depth)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        cum = torch.empty((B, T), dtype=torch.int32, device=dev)
        mel_len = torch.empty((B,), dtype=torch.int32, device=dev)
        # `head` = MelDecoder._head(): also produce the decoder's first stage at phoneme rate (inference only)
        h0 = None if (head is None or train) else torch.empty((B, T, head[0].dx2), dtype=torch.float32, device=dev)
        fp = (C.c_void_p * depth)(*[_ptr(f) for f in feats])
        ni = (C.c_int * depth)(*[f.shape[1] for f in feats])
        # Fuse (channels [0,dim) of feat) + the three predictors, embeddings, concat, duration rounding and the
        # length regulator's scan (cum, mel_len)
        args = (C.byref(fw), depth, dim, self.fuse.kernel_size, B, T, fp, ni, C.byref(pw), C.byref(ew), C.byref(dw), _ptr(m8),
                _ptr(pitch_t), _ptr(energy_t), _ptr(dur_t), _ptr(feat), _ptr(preds[0]), _ptr(preds[1]), _ptr(preds[2]),
                _ptr(idxs[0]), _ptr(idxs[1]), _ptr(dur), _ptr(cum), _ptr(mel_len))
        plan = _lib.current_plan()
        if h0 is not None:
            try:
                lib.esmi_fuse_variance_adaptor_f32(*args, C.byref(head[0]), _ptr(h0), plan, _ptr(ws), ws_bytes, stream)
            except _lib.Unsupported:           # long sequences / other widths: the decoder runs its first stage itself
                h0 = None
        if h0 is None:
            lib.esmi_fuse_variance_adaptor_f32(*args, None, None, plan, _ptr(ws), ws_bytes, stream)
        enc = dict(feat=feat, mask_u8=m8, pitch=preds[0], energy=preds[1], duration=preds[2], pitch_idx=idxs[0],
                   energy_idx=idxs[1], dur=dur, cum=cum, mel_len=mel_len, lmax=None, feats=feats, h0=h0)
        if need_lmax:
            PhonemeEncoder._lmax(enc)
        return enc

    @staticmethod
    def _lmax(enc):
        """Device scalar max_b mel_len[b] (the padded length L); computed on demand: with a caller-supplied output
        length the decoder derives it from mel_len itself and no extra launch is needed."""
        if enc["lmax"] is None:
            lib, stream = _runtime(enc["feat"])
            mel_len = enc["mel_len"]
            enc["lmax"] = torch.empty((1,), dtype=torch.int32, device=mel_len.device)
            lib.esmi_max_i32(_ptr(mel_len), mel_len.shape[0], _ptr(enc["lmax"]), stream)
        return enc["lmax"]

    @staticmethod
    def _padded_len(x, enc, train):
        """(L_out, lmax_dev, lmax_host) for the decoder: L the reference pads to is max(x['mel_len']) when training
        (:344), the batch max otherwise.  `max_mel_len` in x (extension) supplies the allocation length without a
        device->host sync; the exact L then comes from the device (enc['lmax'] if computed, else the decoder derives
        it from mel_len: lmax_host = -1)."""
        if train:
            L = int(torch.max(x["mel_len"]).item())
            return L, None, L
        if "max_mel_len" in x:
            return int(x["max_mel_len"]), enc["lmax"], -1
        L = int(PhonemeEncoder._lmax(enc).item())
        return L, None, L

    def forward(self, x, train=False):
        with _on_device_of(self.encoder.embed.weight):
            return self._forward(x, train)

    def _forward(self, x, train=False):
        enc = self._encode(x, train)
        lib, stream = _runtime(enc["feat"])
        feat, cum, m8 = enc["feat"], enc["cum"], enc["mask_u8"]
        B, T, C4 = feat.shape
        L, _, _ = self._padded_len(x, enc, train)
        features = torch.empty((B, L, C4), dtype=torch.float32, device=feat.device)
        masks8 = torch.empty((B, L), dtype=torch.uint8, device=feat.device)
        if L > 0:
            lib.esmi_upsample_f32(_ptr(feat), _ptr(m8), _ptr(cum), B, T, C4, L, _ptr(features), _ptr(masks8), stream)
        masks = None if m8 is None else masks8.bool().unsqueeze(-1).expand(-1, -1, C4)
        return {"pitch": enc["pitch"], "energy": enc["energy"], "duration": enc["duration"],
                "mel_len": enc["mel_len"], "features": features, "masks": masks}


class Phoneme2Mel(nn.Module):
    """Phoneme sequence -> mel spectrogram (networks.py:404-434)."""

    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder

    def forward(self, x, train=False):
        with _on_device_of(self.decoder.mel_linear.weight):
            return self._forward(x, train)

    def _forward(self, x, train=False):
        if isinstance(x, list):                                       # ONNX-export quirk kept (:418-419)
            x = x[0]
        if train:
            pred = self.encoder(x, train=True)
            mel = self.decoder(pred["features"])
            mask = pred["masks"]
            if mask is not None and mel.size(0) > 1:                  # :424-427
                lib, stream = _runtime(mel)
                m8 = mask[:, :, 0].contiguous().view(torch.uint8)
                lib.esmi_mask_rows_f32(_ptr(mel), _ptr(m8), mel.shape[0] * mel.shape[1], mel.shape[2], stream)
            pred["mel"] = mel
            return pred
        # inference: the (B,L,4*dim) tensor is never materialised -- the decoder gathers through the duration scan and
        # applies the final masked_fill itself; the whole forward is ONE C-ABI call (esmi_phoneme2mel_forward_f32)
        if os.environ.get("ESMI_DEBUG_RANGE") == "1":
            # the FIRST forward of every set of weights runs on the range-checked build (check_activation_range) and raises
            # _lib.ActivationRange where the product build would saturate an activation silently; later forwards of the same
            # weights run the product build.  (Walks the parameters per call: a validation setting, not for serving.)
            key = (_lib.generation(),) + tuple((q.data_ptr(), q._version) for q in self.parameters())
            if getattr(self, "_range_ok_key", None) != key:
                out = self.check_activation_range(x)
                self._range_ok_key = key
                return out
        st = self._launch(x)
        return st.mel, st.mel_len, st.duration

    def check_activation_range(self, x):
        """Validation of a (new) checkpoint on a sample batch: run the inference forward on the range-checked build of the library
        (`libesmi_checked.so`: every value entering a split-f16 contraction is tested against |a| < 65504) and raise
        `_lib.ActivationRange` if one is outside -- the product build would saturate it silently.  Costs a device sync; not for
        serving.  Returns the forward's (mel, mel_len, duration) when everything is in range."""
        w = self.decoder.mel_linear.weight
        checked = os.path.join(os.path.dirname(_lib.LIB_PATH), "libesmi_checked.so")
        ctx = _lib.use_library(checked) if (w.is_cuda and os.path.exists(checked)) else contextlib.nullcontext()
        if w.is_cuda and not os.path.exists(checked):
            raise RuntimeError(f"{checked} not built (python -c 'import __graft_entry__ as g; g.build()')")
        with ctx:
            self._range_flag = torch.zeros(1, dtype=torch.int32, device=w.device)
            try:
                with torch.no_grad():
                    st = self._launch(x)
                return st.mel, st.mel_len, st.duration
            finally:
                self._range_flag = None

    # ------------------------------------------------------------------ one-call forward
    def _static_args(self, lib, stream):
        """The checkpoint-dependent part of esmi_forward_args (packed weights, shapes), rebuilt only when a packed cache was."""
        enc = self.encoder
        blocks, embed = enc.encoder._packed(lib, stream)
        fw, _kf = enc.fuse._packed(lib, stream)
        preds = enc._predictors(lib, stream)
        blob = self.decoder._packed(lib, stream)
        head = self.decoder._head(lib, stream)
        ident = (id(blocks), id(fw), id(preds), id(blob), id(head), id(lib))
        if getattr(self, "_fwd_ident", None) != ident:
            a = _lib.ForwardArgs()
            e = enc.encoder
            a.depth, a.dim, a.fuse_kernel = e.depth, enc.dim, enc.fuse.kernel_size
            for i, (wts, _keep) in enumerate(blocks):
                a.blocks[i] = wts
                a.shapes[i] = _lib.EncoderBlockShape(0, 0, e.dim_ins[i], e.dim_outs[i], e.heads[i], e.kernels[i], e.strides[i],
                                                     e.expansion, N_SYMBOLS + 1, 0, 0, 0)
            a.embed = _ptr(embed)
            a.fuse = fw
            a.pitch, a.energy, a.duration = preds[0][0], preds[1][0], preds[2][0]
            if head is not None:
                a.head = head[0]
            a.dec_blob = _ptr(blob)
            a.dec_shape = self.decoder._shape()
            self._fwd_args, self._fwd_keep, self._fwd_ident = a, (blocks, embed, fw, preds, blob, head), ident
        return self._fwd_args

    def _launch(self, x, stage=0, state=None):
        """Enqueue the inference forward (stage 0), its encoder side only (1), or the decoder (2) on the `state` a stage-1
        call returned (a multi-GPU caller MAX-reduces `state.lmax` in between).  -> namespace with mel, mel_len,
        duration (B,T,1), lmax (device scalar or None)."""
        from types import SimpleNamespace
        w = self.decoder.mel_linear.weight
        lib, stream = _runtime(w)
        dev = w.device
        a = self._static_args(lib, stream)
        if stage == 2:
            st = state
        else:
            phoneme = x["phoneme"]
            B = phoneme.shape[0]
            phoneme_mask = x["phoneme_mask"] if B > 1 else None       # KeyError for B>1, as the reference (:338)
            ids = phoneme.detach().to(device=dev, dtype=torch.int32).contiguous()
            T = ids.shape[1]
            dur_t = None
            if "duration_forced" in x:                                # extension: inject durations at inference
                dur_t = x["duration_forced"].detach().reshape(B, T).to(device=dev, dtype=torch.int32).contiguous()
            st = SimpleNamespace(ids=ids, m8=_mask_u8(phoneme_mask), dur_t=dur_t, B=B, T=T, lmax=None, mel=None, L_out=None,
                                 lmax_host=None, plan=_lib.current_plan(),
                                 mel_len=torch.empty((B,), dtype=torch.int32, device=dev),
                                 duration=torch.empty((B, T, 1), dtype=torch.float32, device=dev))
            if "max_mel_len" in x:                                    # extension: output length from the caller, no host sync
                st.L_out = int(x["max_mel_len"])
                st.lmax_host = st.L_out if bool(x.get("max_mel_len_exact", False)) else -1
            if st.L_out is None or (stage == 1 and st.lmax_host < 0):
                st.lmax = torch.empty((1,), dtype=torch.int32, device=dev)
        # per-call fields of THIS state
        a.B, a.T, a.plan = st.B, st.T, st.plan
        a.ids, a.mask, a.dur_forced = _ptr(st.ids), _ptr(st.m8), _ptr(st.dur_t)
        a.duration_pred, a.mel_len, a.lmax_dev = _ptr(st.duration), _ptr(st.mel_len), _ptr(st.lmax)
        a.pitch_pred = a.energy_pred = a.pitch_idx = a.energy_idx = a.dur = a.cum = None
        if stage != 2:
            a.L_out = st.L_out or 0                                   # known output length: the arena's scratch also fits the decoder's carried rows
            nbytes = lib.esmi_forward_arena_bytes(C.byref(a))
            st.arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        a.arena, a.arena_bytes = _ptr(st.arena), st.arena.numel()
        rf = getattr(self, "_range_flag", None)                       # validation mode only (check_activation_range)
        a.range_flag = _ptr(rf)
        dec = self.decoder
        timed = False
        if stage != 1 and dec.timing is not None:                     # bench.py: HIP events around the decoder launch only
            dec._launches += 1
            timed = dec._launches % dec.timing_every == 0
        if stage != 2 and (stage == 1 or st.L_out is None or timed):  # encoder side as its own call
            a.L_out, a.lmax_host, a.mel = 0, 0, None
            lib.esmi_phoneme2mel_forward_f32(C.byref(a), 1, stream)
            if stage == 1:
                return st
            stage = 2
        if st.L_out is None:                                          # the reference's behaviour: L = batch maximum (host sync)
            st.L_out = st.lmax_host = int(st.lmax.item())
        st.mel = torch.empty((st.B, st.L_out, self.decoder.n_mel_channels), dtype=torch.float32, device=dev)
        a.mel, a.L_out, a.lmax_host = _ptr(st.mel), st.L_out, st.lmax_host
        if timed and st.L_out > 0:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            lib.esmi_phoneme2mel_forward_f32(C.byref(a), stage, stream)
            ev[1].record()
            dec.timing.append(ev)
        else:
            lib.esmi_phoneme2mel_forward_f32(C.byref(a), stage, stream)
        return st
