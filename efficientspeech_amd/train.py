"""Training step of the acoustic model on MI355X (reference: model.py:167-226 loss + training_step,
:279-283 AdamW, train.py:66-76; the train=True data flow of layers/networks.py:336-434).

Every arithmetic operation of the step is a HIP kernel behind the C-ABI (`esmi_train_*`, csrc/train_ops.h): forward operators
that keep what their backward reads, data- and weight-gradient kernels, the fused masked loss, AdamW.  torch.autograd is used
as the TAPE only (which operator's backward runs when, and the accumulation of gradients that fan in); torch.distributed
(RCCL) carries the one gradient all-reduce.  Parameters and gradients of the model live in ONE flat fp32 buffer each
(`FlatParams`): the data-parallel exchange is a single all-reduce of 1.07 MB (tiny ES) and the optimizer a single launch.

The operators take the model's own `nn.Parameter`s in checkpoint layout, so `Phoneme2Mel.state_dict()` after N steps is a
reference-compatible checkpoint and the inference path (`networks.py`) picks the updated weights up through its pack cache.
"""
import ctypes as C

import torch

from . import _lib, networks
from .networks import _ptr, _mask_u8

ACT_RELU, ACT_GELU, ACT_TANH = 1, 2, 3
USE_MATRIX_PIPE = True    # dense convolutions (forward and data gradient) through the implicit-GEMM kernels; False: plain fp32 kernels
USE_MATRIX_PIPE_DGRAD = True   # (development) False: only the forward GEMMs on the matrix pipe
PRECISION = 32            # 16: the reference's `--precision 16` -- the dense GEMMs (forward, data gradient) round their operands to
#                           binary16 (one MFMA product instead of three); set by TrainStep(precision=16) around its step


def _rt(t):
    return networks._runtime(t)


def _new(shape, like, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


_REDUCE_Q = None          # (esmi_reduce_queue, [workspaces kept alive]) while TrainStep's backward runs: the second stages of the chunked
#                           parameter-gradient reductions are queued and run as ONE launch before the optimizer (62 launches otherwise)


def _defer(ws, *direct):
    """The queue to hand to an operator's backward, or None: only gradients that go straight into the flat buffer may be late
    (autograd would read a returned tensor at once); the partial sums in `ws` must outlive the flush."""
    if _REDUCE_Q is None or not all(direct):
        return None
    _REDUCE_Q[1].append(ws)
    return C.byref(_REDUCE_Q[0])


_DIRECT_GRADS = False     # True only while TrainStep._body runs: ONE backward per zeroed buffer, so overwriting == accumulating
_LOSS_SEED = None         # while TrainStep's step runs: False = the backward is seeded with 1, or the device scalar it is seeded with (the loss
#                           scale): the loss kernel writes its gradients already multiplied by it and `_Loss.backward` passes them on as they are
_PACKED_VALID = False     # True only while TrainStep._fwd_bwd runs, after its pack launch: `param._esmi_packed` holds this step's GEMM copies


def _grad_buffer(param):
    """Where an operator's backward writes a PARAMETER's gradient: straight into the flat gradient buffer when `FlatParams` owns
    the parameter AND a `TrainStep` is driving the step (then autograd gets None for it: no temporary, no accumulation launch),
    else a fresh tensor for autograd to accumulate -- so `training_step(...).backward()` called twice before an optimizer step
    (gradient accumulation through the wrapper's public route) adds up as torch semantics require.  The direct write is valid
    inside `TrainStep` because the buffer is zeroed per step and every parameter of this model feeds exactly one operator."""
    view = getattr(param, "_esmi_grad_view", None)
    if _DIRECT_GRADS and view is not None and param.grad is not None and param.grad.data_ptr() == view.data_ptr():   # still the buffer autograd would add to
        return view, True
    return torch.empty_like(param), False


# --------------------------------------------------------------------------- operators (autograd = the tape)
class _Conv(torch.autograd.Function):
    """Conv1d / ConvTranspose1d / Linear on channels-last (B, n, C); `w` in checkpoint layout (Linear: (Cout, Cin))."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, groups, transposed, n_out, act=0, act_grad_downstream=False):
        """act: ReLU / tanh applied to the result inside the convolution's launch (`esmi_conv_desc.act`); their derivatives come
        from the saved output, so the backward is act' then the convolution's -- unless the one consumer of the result takes care of
        act' in its own backward (`act_grad_downstream`: the LayerNorm that follows, `in_act`)."""
        assert act in (0, ACT_RELU, ACT_TANH)
        ctx.act_here = bool(act) and not act_grad_downstream
        w0 = w
        x, w = x.contiguous(), w.contiguous()
        lib, st = _rt(x)
        B, n_in, c_in = x.shape
        w3 = w if w.dim() == 3 else w.unsqueeze(-1)
        c_out, k = (w3.shape[1] if transposed else w3.shape[0]), w3.shape[2]
        d = _lib.ConvDesc(B, n_in, c_in, n_out, c_out, k, stride, pad, groups, 1 if transposed else 0, 16 if PRECISION == 16 else 0, act)
        packed = getattr(w0, "_esmi_packed", None) if (_PACKED_VALID and USE_MATRIX_PIPE) else None
        if packed is not None:                  # this step's GEMM copies, made by TrainStep's one pack launch
            d.packed_fwd, d.packed_grad = _ptr(packed[0]), (_ptr(packed[1]) if USE_MATRIX_PIPE_DGRAD else None)
        y = _new((B, n_out, c_out), x)
        nws = lib.esmi_train_conv_workspace_bytes(C.byref(d)) if (USE_MATRIX_PIPE and packed is None) else 0
        ws = _new((nws,), x, torch.uint8) if nws else None
        lib.esmi_train_conv_fwd_f32(C.byref(d), _ptr(x), _ptr(w), _ptr(b), _ptr(y), _ptr(ws), nws, st)
        ctx.save_for_backward(x, w, *((y,) if ctx.act_here else ()))
        ctx.d, ctx.params = d, (w0, b)          # the Parameter objects themselves: their flat gradient views are the outputs
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors[:2]
        dy = dy.contiguous()
        lib, st = _rt(dy)
        d = ctx.d
        if ctx.act_here:
            da = torch.empty_like(dy)
            lib.esmi_train_act_bwd_f32(_ptr(ctx.saved_tensors[2]), _ptr(dy), dy.numel(), d.act, _ptr(da), st)
            dy = da
        dx = torch.empty_like(x)
        w0, b0 = ctx.params
        dw, w_direct = _grad_buffer(w0)
        db, b_direct = _grad_buffer(b0) if b0 is not None else (None, True)
        if USE_MATRIX_PIPE and USE_MATRIX_PIPE_DGRAD:
            # one call for both gradients: the weight-gradient pass leaves max|dy| behind for the data-gradient GEMM's operand scale
            nws = lib.esmi_train_conv_bwd_workspace_bytes(C.byref(d))
            ws = _new((nws,), w, torch.uint8)
            lib.esmi_train_conv_bwd_f32(C.byref(d), _ptr(x), _ptr(dy), _ptr(w), _ptr(dx), _ptr(dw), _ptr(db), _ptr(ws), nws,
                                        _defer(ws, w_direct, b_direct), st)
        else:
            lib.esmi_train_conv_dgrad_f32(C.byref(d), _ptr(dy), _ptr(w), _ptr(dx), None, 0, st)
            nws = lib.esmi_train_conv_wgrad_workspace_bytes(C.byref(d))
            ws = _new((nws,), w, torch.uint8)
            lib.esmi_train_conv_wgrad_f32(C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), nws, st)
        return dx, (None if w_direct else dw), (None if b_direct else db), None, None, None, None, None, None, None


class _LayerNorm(torch.autograd.Function):
    """LayerNorm over the last dim -- optionally of x + res (both summands get the backward's dx) and with padded rows zeroed
    afterwards (`mask` (rows) uint8; those rows pass no gradient): the add and the masked_fill ride in the norm's launches.
    in_act: x is the output of that ReLU / tanh and has no other consumer -- the backward returns the gradient of the activation's
    INPUT, so the producer (`conv(..., act=kind, act_grad_downstream=True)`) must not apply the activation's backward again."""

    @staticmethod
    def forward(ctx, x, g, b, res=None, mask=None, in_act=0, relu_out=False):
        """relu_out: y = relu(LN(.)) in the same launch (networks.py:153-154); the backward gates dy with the saved output."""
        assert in_act in (0, ACT_RELU, ACT_TANH) and not (in_act and res is not None)
        x = x.contiguous()
        lib, st = _rt(x)
        rows, Cc = x.numel() // x.shape[-1], x.shape[-1]
        y, mean, rstd = torch.empty_like(x), _new((rows,), x), _new((rows,), x)
        xs = x
        if res is not None:
            res, xs = res.contiguous(), torch.empty_like(x)
        lib.esmi_train_layernorm_fwd_f32(_ptr(x), _ptr(g), _ptr(b), rows, Cc, _ptr(y), _ptr(mean), _ptr(rstd), _ptr(res),
                                         _ptr(xs) if res is not None else None, _ptr(mask), 1 if relu_out else 0, st)
        ctx.save_for_backward(xs, g, mean, rstd, mask, y if relu_out else None)
        ctx.params, ctx.has_res, ctx.in_act = (g, b), res is not None, in_act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, mean, rstd, mask, y_relu = ctx.saved_tensors
        dy = dy.contiguous()
        lib, st = _rt(dy)
        rows, Cc = x.numel() // x.shape[-1], x.shape[-1]
        dx = torch.empty_like(x)
        (dg, g_direct), (db, b_direct) = _grad_buffer(ctx.params[0]), _grad_buffer(ctx.params[1])
        nws = lib.esmi_train_layernorm_bwd_workspace_bytes(rows, Cc)
        ws = _new((nws,), x, torch.uint8)
        lib.esmi_train_layernorm_bwd_f32(_ptr(x), _ptr(g), _ptr(mean), _ptr(rstd), _ptr(dy), rows, Cc, _ptr(dx), _ptr(dg), _ptr(db),
                                         _ptr(ws), nws, _defer(ws, g_direct, b_direct), _ptr(mask), ctx.in_act, _ptr(y_relu), st)
        return dx, (None if g_direct else dg), (None if b_direct else db), (dx if ctx.has_res else None), None, None, None


class _ConvLN(torch.autograd.Function):
    """conv (stride 1) / Linear, its activation, an optional residual add, LayerNorm, the norm's ReLU and row mask in ONE forward
    launch (`esmi_train_conv_ln_fwd_f32`: the norm in the GEMM's epilogue) -- the chain `layer_norm(conv(x, m, act, True), norm, res,
    mask, in_act=act, relu_out)` of two launches otherwise.  The backward is that chain's: the fused LayerNorm backward (which also
    runs the activation's), then the convolution's two gradients.  Shapes the fused launch does not take run the two forward launches
    here, so the saved tensors and the backward are the same either way."""

    @staticmethod
    def forward(ctx, x, w, b, pad, groups, act, g, beta, res, mask, relu_out):
        assert act in (0, ACT_RELU, ACT_TANH) and not (act and res is not None)
        w0 = w
        x, w = x.contiguous(), w.contiguous()
        lib, st = _rt(x)
        B, n, c_in = x.shape
        w3 = w if w.dim() == 3 else w.unsqueeze(-1)
        c_out, k = w3.shape[0], w3.shape[2]
        d = _lib.ConvDesc(B, n, c_in, n, c_out, k, 1, pad, groups, 0, 16 if PRECISION == 16 else 0, act)
        packed = getattr(w0, "_esmi_packed", None) if (_PACKED_VALID and USE_MATRIX_PIPE) else None
        if packed is not None:
            d.packed_fwd, d.packed_grad = _ptr(packed[0]), (_ptr(packed[1]) if USE_MATRIX_PIPE_DGRAD else None)
        rows = B * n
        y_pre, y, mean, rstd = _new((B, n, c_out), x), _new((B, n, c_out), x), _new((rows,), x), _new((rows,), x)
        nws = lib.esmi_train_conv_workspace_bytes(C.byref(d)) if (USE_MATRIX_PIPE and packed is None) else 0
        ws = _new((nws,), x, torch.uint8) if nws else None
        if res is not None:
            res = res.contiguous()
        fused = False
        if USE_MATRIX_PIPE and groups == 1:
            try:
                lib.esmi_train_conv_ln_fwd_f32(C.byref(d), _ptr(x), _ptr(w), _ptr(b), _ptr(res), _ptr(g), _ptr(beta), _ptr(mask),
                                               1 if relu_out else 0, _ptr(y_pre), _ptr(y), _ptr(mean), _ptr(rstd), _ptr(ws), nws, st)
                fused = True
            except _lib.Unsupported:
                pass
        if not fused:                       # the two launches (conv into y, then the norm in place: it reads a row before it writes it)
            tmp = y if res is not None else y_pre
            lib.esmi_train_conv_fwd_f32(C.byref(d), _ptr(x), _ptr(w), _ptr(b), _ptr(tmp), _ptr(ws), nws, st)
            lib.esmi_train_layernorm_fwd_f32(_ptr(tmp), _ptr(g), _ptr(beta), rows, c_out, _ptr(y), _ptr(mean), _ptr(rstd),
                                             _ptr(res), _ptr(y_pre) if res is not None else None, _ptr(mask), 1 if relu_out else 0, st)
        ctx.save_for_backward(x, w, y_pre, g, mean, rstd, mask, y if relu_out else None)
        ctx.d, ctx.params, ctx.has_res, ctx.act = d, (w0, b, g, beta), res is not None, act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y_pre, g, mean, rstd, mask, y_relu = ctx.saved_tensors
        dy = dy.contiguous()
        lib, st = _rt(dy)
        d = ctx.d
        rows, Cc = y_pre.numel() // y_pre.shape[-1], y_pre.shape[-1]
        w0, b0, g0, beta0 = ctx.params
        dpre = torch.empty_like(y_pre)
        (dg, g_direct), (dbt, bt_direct) = _grad_buffer(g0), _grad_buffer(beta0)
        nws = lib.esmi_train_layernorm_bwd_workspace_bytes(rows, Cc)
        ws = _new((nws,), x, torch.uint8)
        lib.esmi_train_layernorm_bwd_f32(_ptr(y_pre), _ptr(g), _ptr(mean), _ptr(rstd), _ptr(dy), rows, Cc, _ptr(dpre), _ptr(dg), _ptr(dbt),
                                         _ptr(ws), nws, _defer(ws, g_direct, bt_direct), _ptr(mask), ctx.act, _ptr(y_relu), st)
        dx = torch.empty_like(x)
        dw, w_direct = _grad_buffer(w0)
        db, b_direct = _grad_buffer(b0) if b0 is not None else (None, True)
        if USE_MATRIX_PIPE and USE_MATRIX_PIPE_DGRAD:
            nws = lib.esmi_train_conv_bwd_workspace_bytes(C.byref(d))
            ws = _new((nws,), w, torch.uint8)
            lib.esmi_train_conv_bwd_f32(C.byref(d), _ptr(x), _ptr(dpre), _ptr(w), _ptr(dx), _ptr(dw), _ptr(db), _ptr(ws), nws,
                                        _defer(ws, w_direct, b_direct), st)
        else:
            lib.esmi_train_conv_dgrad_f32(C.byref(d), _ptr(dpre), _ptr(w), _ptr(dx), None, 0, st)
            nws = lib.esmi_train_conv_wgrad_workspace_bytes(C.byref(d))
            ws = _new((nws,), w, torch.uint8)
            lib.esmi_train_conv_wgrad_f32(C.byref(d), _ptr(x), _ptr(dpre), _ptr(dw), _ptr(db), _ptr(ws), nws, st)
        return (dx, (None if w_direct else dw), (None if b_direct else db), None, None, None, (None if g_direct else dg),
                (None if bt_direct else dbt), (dpre if ctx.has_res else None), None, None)


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind):
        x = x.contiguous()
        lib, st = _rt(x)
        y = torch.empty_like(x)
        lib.esmi_train_act_fwd_f32(_ptr(x), x.numel(), kind, _ptr(y), st)
        ctx.save_for_backward(x if kind == ACT_GELU else y)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        (saved,) = ctx.saved_tensors
        dy = dy.contiguous()
        lib, st = _rt(dy)
        dx = torch.empty_like(dy)
        lib.esmi_train_act_bwd_f32(_ptr(saved), _ptr(dy), dy.numel(), ctx.kind, _ptr(dx), st)
        return dx, None


class _AttnCore(torch.autograd.Function):
    """qkv (B, N, 3*h*C) as the qkv Linear leaves it -> softmax(q k^T / sqrt(C/h)) v, heads concatenated (B, N, h*C)."""

    @staticmethod
    def forward(ctx, qkv, h):
        qkv = qkv.contiguous()
        lib, st = _rt(qkv)
        B, N, w = qkv.shape
        Cc = w // (3 * h)
        P, out = _new((B, h, N, N), qkv), _new((B, N, h * Cc), qkv)
        lib.esmi_train_attention_fwd_f32(_ptr(qkv), B, N, Cc, h, _ptr(P), _ptr(out), st)
        ctx.save_for_backward(qkv, P)
        ctx.dims = (B, N, Cc, h)
        return out

    @staticmethod
    def backward(ctx, dctx):
        qkv, P = ctx.saved_tensors
        dctx = dctx.contiguous()
        lib, st = _rt(dctx)
        B, N, Cc, h = ctx.dims
        dS, dqkv = torch.empty_like(P), torch.empty_like(qkv)
        lib.esmi_train_attention_bwd_f32(_ptr(qkv), _ptr(P), _ptr(dctx), B, N, Cc, h, _ptr(dS), _ptr(dqkv), st)
        return dqkv, None


class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, table, padding_idx):
        ids = ids.contiguous().to(torch.int32)
        lib, st = _rt(table)
        V, Cc = table.shape
        out = _new(tuple(ids.shape) + (Cc,), table)
        lib.esmi_train_embedding_fwd_f32(_ptr(ids), _ptr(table), ids.numel(), V, Cc, _ptr(out), st)
        ctx.save_for_backward(ids)
        ctx.dims, ctx.param = (V, Cc, padding_idx), table
        return out

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        dy = dy.contiguous()
        lib, st = _rt(dy)
        V, Cc, pad = ctx.dims
        dt, direct = _grad_buffer(ctx.param)
        nws = lib.esmi_train_embedding_bwd_workspace_bytes(ids.numel(), V, Cc)
        ws = _new((nws,), dy, torch.uint8)
        lib.esmi_train_embedding_bwd_f32(_ptr(ids), _ptr(dy), ids.numel(), V, Cc, pad, _ptr(dt), _ptr(ws), nws, _defer(ws, direct), st)
        return None, (None if direct else dt), None


class _MaskRows(torch.autograd.Function):
    """x.masked_fill(mask[..., None], 0) for a (B, n) uint8 mask."""

    @staticmethod
    def forward(ctx, x, mask_u8):
        x = x.contiguous()
        lib, st = _rt(x)
        y = torch.empty_like(x)
        lib.esmi_train_mask_rows_f32(_ptr(x), _ptr(mask_u8), x.numel() // x.shape[-1], x.shape[-1], _ptr(y), st)
        ctx.save_for_backward(mask_u8)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask_u8,) = ctx.saved_tensors
        dy = dy.contiguous()
        lib, st = _rt(dy)
        dx = torch.empty_like(dy)
        lib.esmi_train_mask_rows_f32(_ptr(dy), _ptr(mask_u8), dy.numel() // dy.shape[-1], dy.shape[-1], _ptr(dx), st)
        return dx, None


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        lib, st = _rt(a)
        y = torch.empty_like(a)
        lib.esmi_train_add_f32(_ptr(a), _ptr(b), a.numel(), _ptr(y), st)
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class _Cat(torch.autograd.Function):
    """torch.cat(parts, dim=-1) in one launch; parts whose bit is set in `masked` are first zeroed on the rows of the (rows) uint8
    `mask` (x.masked_fill(mask[..., None], 0) in front of the cat), gradients likewise."""

    @staticmethod
    def _run(lib, st, parts, widths, rows, cat, mask, masked, backward):
        ptrs = (C.c_void_p * len(parts))(*[_ptr(p) for p in parts])
        lib.esmi_train_cat_f32(ptrs, (C.c_int * len(parts))(*widths), len(parts), rows, _ptr(cat), _ptr(mask), masked, backward, st)

    @staticmethod
    def forward(ctx, mask, masked, *parts):
        parts = [p.contiguous() for p in parts]
        lib, st = _rt(parts[0])
        widths = [p.shape[-1] for p in parts]
        rows, tot = parts[0].numel() // widths[0], sum(widths)
        y = _new(tuple(parts[0].shape[:-1]) + (tot,), parts[0])
        masked = masked if mask is not None else 0
        _Cat._run(lib, st, parts, widths, rows, y, mask, masked, 0)
        ctx.widths, ctx.masked = widths, masked
        ctx.save_for_backward(*((mask,) if masked else ()))
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        lib, st = _rt(dy)
        rows = dy.numel() // dy.shape[-1]
        outs = [_new(tuple(dy.shape[:-1]) + (w,), dy) for w in ctx.widths]
        _Cat._run(lib, st, outs, ctx.widths, rows, dy, ctx.saved_tensors[0] if ctx.masked else None, ctx.masked, 1)
        return (None, None) + tuple(outs)


class _Repeat(torch.autograd.Function):
    """FeatureUpsampler: (B, T, C) -> (B, L, C) by the inclusive duration cumsum `cum` (B, T) int32."""

    @staticmethod
    def forward(ctx, feat, cum, L):
        feat = feat.contiguous()
        lib, st = _rt(feat)
        B, T, Cc = feat.shape
        out = _new((B, L, Cc), feat)
        lib.esmi_train_repeat_fwd_f32(_ptr(feat), _ptr(cum), B, T, Cc, L, _ptr(out), st)
        ctx.save_for_backward(cum)
        ctx.dims = (B, T, Cc, L)
        return out

    @staticmethod
    def backward(ctx, dout):
        (cum,) = ctx.saved_tensors
        dout = dout.contiguous()
        lib, st = _rt(dout)
        B, T, Cc, L = ctx.dims
        df = _new((B, T, Cc), dout)
        lib.esmi_train_repeat_bwd_f32(_ptr(dout), _ptr(cum), B, T, Cc, L, _ptr(df), st)
        return df, None, None


class _Loss(torch.autograd.Function):
    """model.py:167-216.  Returns (parts, total): the four means (mel L1, pitch MSE, energy MSE, log-duration MSE; reported, not
    differentiable -- the reference's `loss()` values are only ever logged) and the weighted total, whose backward hands the
    kernel's gradient seeds d total / d prediction to the four predictions, scaled by the incoming seed."""

    @staticmethod
    def forward(ctx, mel_pred, pitch_pred, energy_pred, dur_pred, mel, pitch, energy, dur, mel_mask, ph_mask):
        mel_pred, pitch_pred, energy_pred, dur_pred = (t.contiguous() for t in (mel_pred, pitch_pred, energy_pred, dur_pred))
        lib, st = _rt(mel_pred)
        B, L, nm = mel_pred.shape
        T = pitch_pred.shape[1]
        out = _new((5,), mel_pred)
        grads = [torch.empty_like(t) for t in (mel_pred, pitch_pred, energy_pred, dur_pred)]
        scratch = _new((1536,), mel_pred)          # ESMI_TRAIN_LOSS_SCRATCH_FLOATS; held until the call has been enqueued
        seed = _LOSS_SEED                          # None: unknown here (scaled in backward); False: 1; else the device scalar
        a = _lib.TrainLossArgs(_ptr(mel_pred), _ptr(mel), _ptr(pitch_pred), _ptr(pitch), _ptr(energy_pred), _ptr(energy),
                               _ptr(dur_pred), _ptr(dur), _ptr(mel_mask), _ptr(ph_mask), B, T, L, nm, _ptr(out),
                               *[_ptr(g) for g in grads], _ptr(scratch), _ptr(seed) if torch.is_tensor(seed) else None)
        lib.esmi_train_loss_f32(C.byref(a), st)
        ctx.save_for_backward(*grads)
        ctx.prescaled = seed is not None
        parts, total = out[:4], out[4]
        ctx.mark_non_differentiable(parts)
        return parts, total

    @staticmethod
    def backward(ctx, _dparts, dtotal):
        if ctx.prescaled:                          # TrainStep told the forward what this backward is seeded with
            return tuple(ctx.saved_tensors) + (None,) * 6
        return tuple(g * dtotal for g in ctx.saved_tensors) + (None,) * 6     # (seed 1 from `.backward()`: four scalings, exact)


def loss_vector(parts, total):
    """The 5-vector (mel, pitch, energy, duration, total) for reporting."""
    return torch.cat([parts.detach(), total.detach().reshape(1)])


def conv(x, m, n_out=None, act=0, act_grad_downstream=False):
    """Apply an nn.Conv1d / nn.ConvTranspose1d / nn.Linear parameter container to channels-last x (+ ReLU / tanh in the same launch)."""
    if isinstance(m, torch.nn.Linear):
        return _Conv.apply(x, m.weight, m.bias, 1, 0, 1, False, x.shape[1], act, act_grad_downstream)
    tr = isinstance(m, torch.nn.ConvTranspose1d)
    k, s, p = m.kernel_size[0], m.stride[0], m.padding[0]
    full = (x.shape[1] - 1) * s - 2 * p + k if tr else (x.shape[1] + 2 * p - k) // s + 1
    return _Conv.apply(x, m.weight, m.bias, s, p, m.groups, tr, full if n_out is None else min(full, n_out), act, act_grad_downstream)


def layer_norm(x, m, res=None, mask=None, in_act=0, relu_out=False):
    return _LayerNorm.apply(x, m.weight, m.bias, res, mask, in_act, relu_out)


FUSE_CONV_LN = False      # True: a convolution and the LayerNorm behind it are ONE forward launch (`_ConvLN`, esmi_train_conv_ln_fwd_f32).
#                           Measured on tiny ES at B = 128 (profiles/r05_probes/train_conv_ln_fusion.md): 194 -> 182 launches, but 3.14 -> 3.19 ms of
#                           kernels per step -- the norm in the GEMM's epilogue (two row reductions per row in the MFMA C layout, a second
#                           full-size store for the pre-norm tensor) costs more than the 15 us stand-alone norm it replaces: off by default


def conv_ln(x, m, norm, act=0, res=None, mask=None, relu_out=False):
    """layer_norm(conv(x, m, act, act_grad_downstream=True), norm, res, mask, in_act=act, relu_out); with FUSE_CONV_LN the norm rides in
    the convolution's launch (`_ConvLN`).  m: a Linear or a stride-1 Conv1d."""
    if not FUSE_CONV_LN:
        return layer_norm(conv(x, m, act=act, act_grad_downstream=bool(act)), norm, res=res, mask=mask, in_act=act, relu_out=relu_out)
    if isinstance(m, torch.nn.Linear):
        return _ConvLN.apply(x, m.weight, m.bias, 0, 1, act, norm.weight, norm.bias, res, mask, relu_out)
    assert not isinstance(m, torch.nn.ConvTranspose1d) and m.stride[0] == 1 and x.shape[1] + 2 * m.padding[0] - (m.kernel_size[0] - 1) == x.shape[1]
    return _ConvLN.apply(x, m.weight, m.bias, m.padding[0], m.groups, act, norm.weight, norm.bias, res, mask, relu_out)


def act(x, kind):
    return _Act.apply(x, kind)


# --------------------------------------------------------------------------- the train=True forward, operator by operator
def _pooled_mask(mask_u8, n, T, n_out):
    """blocks.py:51-57 for the pool factor the encoder derives (networks.py:71-72)."""
    pool = int(torch.round(torch.tensor([n / n_out])).item())
    if pool <= 1:
        return mask_u8
    lib, st = _rt(mask_u8)
    out = _new((mask_u8.shape[0], n_out), mask_u8, torch.uint8)
    lib.esmi_pool_mask_u8(_ptr(mask_u8), mask_u8.shape[0], T, pool, _ptr(out), n_out, st)
    return out


def encoder_forward(enc, phoneme, mask_u8):
    """Encoder.forward, networks.py:52-87."""
    x = _Embedding.apply(phoneme, enc.embed.weight, 0)
    T = x.shape[1]
    feats = []
    for merge3, merge1, attn, ffn, norm1, norm2 in enc.attn_blocks:
        x = conv(conv(x, merge3), merge1)
        m = _pooled_mask(mask_u8, T, T, x.shape[1]) if mask_u8 is not None else None
        # LN(attn(x) + x) and LN(ffn(x) + x), padded positions zeroed: the add, the norm and the mask ride in the last Linear's launch
        x = conv_ln(_AttnCore.apply(conv(x, attn.qkv), attn.num_heads), attn.proj, norm1, res=x, mask=m)
        x = conv_ln(act(conv(conv(x, ffn.mlp1), ffn.conv), ACT_GELU), ffn.mlp2, norm2, res=x, mask=m)
        feats.append(x)
    return feats


def fuse_forward(fuse, feats, mask_u8):
    """Fuse.forward, networks.py:196-219."""
    T = feats[0].shape[1]
    parts = []
    for f, (mlp, up) in zip(feats, fuse.mlps):
        x = conv(f, mlp)
        if isinstance(up, torch.nn.ConvTranspose1d):
            x = conv(x, up, n_out=T)
        parts.append(x)
    x = conv(_Cat.apply(None, 0, *parts), fuse.fuse)
    return _MaskRows.apply(x, mask_u8) if mask_u8 is not None else x


def predictor_forward(dec, fused):
    """AcousticDecoder.forward, networks.py:151-165 -> (pred (B, T, 1), features (B, T, dim))."""
    y = conv_ln(fused, dec.conv1[0], dec.norm1, act=ACT_RELU, relu_out=True)
    y = conv(y, dec.conv2[0], act=ACT_RELU)
    if dec.duration:
        return conv(y, dec.linear, act=ACT_RELU), layer_norm(y, dec.norm2)
    pred = conv(y, dec.linear)
    return pred, None


def _bucket_embedding(dec, target):
    """get_embedding with a target (networks.py:128-149): the table row of torch.bucketize(target, bins)."""
    bins = dec.pitch_bins if dec.pitch_bins is not None else dec.energy_bins
    emb = dec.pitch_embedding if dec.pitch_embedding is not None else dec.energy_embedding
    lib, st = _rt(target)
    t = target.contiguous().float()
    idx = _new(tuple(t.shape), t, torch.int32)
    scratch = _new(tuple(t.shape) + (emb.weight.shape[1],), t)
    lib.esmi_bucket_embedding_f32(_ptr(t), _ptr(bins), _ptr(emb.weight.detach()), t.numel(), emb.weight.shape[1], _ptr(scratch), _ptr(idx), st)
    return _Embedding.apply(idx, emb.weight, -1)


def decoder_forward(dec, features):
    """MelDecoder.forward, networks.py:291-304."""
    skip = conv_ln(features, dec.proj[0], dec.proj[2], act=ACT_TANH)
    for convs, skip_norm in dec.blocks:
        x = skip
        for seq, norm in convs:
            x = conv_ln(conv(x, seq[0]), seq[1], norm, act=ACT_TANH)
        skip = layer_norm(x, skip_norm, res=skip)
    return conv(skip, dec.mel_linear)


def train_forward(net, x, mask_mel=True):
    """Phoneme2Mel.forward(x, train=True), networks.py:336-434 -> dict(mel, pitch, energy, duration, mel_len)."""
    pe = net.encoder
    phoneme = x["phoneme"]
    B, T = phoneme.shape
    ph_mask = _mask_u8(x["phoneme_mask"]) if B > 1 else None
    feats = encoder_forward(pe.encoder, phoneme, ph_mask)
    fused = fuse_forward(pe.fuse, feats, ph_mask)
    pitch_pred, _ = predictor_forward(pe.pitch_decoder, fused)
    energy_pred, _ = predictor_forward(pe.energy_decoder, fused)
    dur_pred, dur_feat = predictor_forward(pe.duration_decoder, fused)
    pf, ef = _bucket_embedding(pe.pitch_decoder, x["pitch"]), _bucket_embedding(pe.energy_decoder, x["energy"])
    feat4 = _Cat.apply(ph_mask, 0b1110, fused, pf, ef, dur_feat)     # the three masked_fills (networks.py:366-368) inside the cat's launch
    # length regulator on the TARGET durations (masked, clamped at 0), padded to the batch's longest target mel
    lib, st = _rt(feat4)
    dur = x["duration"].to(torch.int32).contiguous()
    if ph_mask is not None:
        dur = dur.masked_fill(x["phoneme_mask"], 0)
    cum, mel_len, lmax = _new((B, T), feat4, torch.int32), _new((B,), feat4, torch.int32), _new((1,), feat4, torch.int32)
    lib.esmi_length_regulate_i32(_ptr(dur), B, T, _ptr(cum), _ptr(mel_len), _ptr(lmax), st)
    L = int(x["mel"].shape[1]) if "mel" in x else int(torch.max(x["mel_len"]).item())
    features = _Repeat.apply(feat4, cum, L)
    mel = decoder_forward(net.decoder, features)
    if ph_mask is not None and mask_mel:
        frames = torch.arange(L, device=mel.device)[None, :] >= mel_len[:, None]        # FeatureUpsampler's masks, one bit per frame
        mel = _MaskRows.apply(mel, _mask_u8(frames))
    return {"mel": mel, "pitch": pitch_pred, "energy": energy_pred, "duration": dur_pred, "mel_len": mel_len}


def training_loss(net, x, y):
    """training_step's forward + loss (model.py:212-216): returns (parts, total) -- the four reported means and the differentiable
    weighted total 10 mel + 2 pitch + 2 energy + duration."""
    xx = dict(x)
    xx.setdefault("mel", y["mel"])
    # (the masked_fill of the padded mel frames, networks.py:423-424, is skipped here: the loss leaves those frames out and
    #  gives them a zero gradient either way)
    out = train_forward(net, xx, mask_mel=False)
    B, T = x["phoneme"].shape
    ph_mask = _mask_u8(x["phoneme_mask"]) if x.get("phoneme_mask") is not None else None
    mel_mask = _mask_u8(x["mel_mask"]) if x.get("mel_mask") is not None else None
    f = lambda t: t.contiguous().float()      # noqa: E731
    return _Loss.apply(out["mel"], out["pitch"].reshape(B, T), out["energy"].reshape(B, T), out["duration"].reshape(B, T),
                       f(y["mel"]), f(x["pitch"]), f(x["energy"]), x["duration"].to(torch.int32).contiguous(), mel_mask, ph_mask)


# --------------------------------------------------------------------------- flat parameters, AdamW, data-parallel step
# parameters the train=True graph never reaches (their .grad stays None in the reference, so AdamW skips them): the pitch /
# energy predictors' norm2 feeds only `features`, which only the duration predictor returns (networks.py:160-165, 347-365)
_UNREACHED = ("encoder.pitch_decoder.norm2.", "encoder.energy_decoder.norm2.")


class FlatParams:
    """All trainable parameters of a module as views into one flat fp32 buffer, their gradients into another."""

    def __init__(self, net):
        named = [(k, p) for k, p in net.named_parameters() if p.requires_grad and not k.startswith(_UNREACHED)]
        self.names = [k for k, _ in named]
        pad4 = lambda k: (k + 3) & ~3                                     # noqa: E731  every tensor starts 16-byte aligned
        n = sum(pad4(p.numel()) for _, p in named)
        dev = named[0][1].device
        self.data = torch.zeros(n, dtype=torch.float32, device=dev)          # (pad elements stay 0 under AdamW: g = 0, p = 0)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m, self.v = torch.zeros_like(self.data), torch.zeros_like(self.data)
        off = 0
        for _, p in named:
            k = p.numel()
            self.data[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.data[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
            p._esmi_grad_view = p.grad              # operators' backward passes write here directly (see _grad_buffer)
            off += pad4(k)
        self.params = [p for _, p in named]

    def zero_grad(self):
        self.grad.zero_()


class TrainStep:
    """One optimizer step of the reference's training loop (model.py:212-226 + :279-283) on one GPU of a data-parallel job.

    step(x, y): forward + loss + backward on this rank's batch, ONE all-reduce (mean) of the flat gradient buffer over `group`
    (RCCL when the process group is `nccl`), one AdamW launch.  lr follows torch.optim.AdamW's defaults as model.py sets them.

    graph=True captures the step into one hipGraph the first time a batch SHAPE is seen and replays it for every later batch of
    that shape (inputs are copied into the graph's static buffers; step count, learning rate and the loss scaler live in device
    memory): the whole step on one GPU, everything up to the gradient all-reduce when data-parallel (the collective and the
    optimizer launch then follow eagerly).  With ~330 launches per step the eager step is bound by the host's dispatch (5.6 ms at
    B = 128); the replay runs at the kernels' own pace (4.9 ms)."""

    def __init__(self, net, lr=1e-3, weight_decay=1e-6, betas=(0.9, 0.999), eps=1e-8, group=None, world_size=1, graph=False,
                 precision=32, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.net, self.flat = net, FlatParams(net)
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.group, self.world = group, world_size
        self._t = 0
        # precision 16 = what Lightning's `precision=16` does around the reference's step (train.py:66-70): autocast-class GEMMs
        # (binary16 operands, fp32 accumulate, fp32 master weights) + torch.amp.GradScaler's dynamic loss scaling: the backward
        # is seeded with `scale`, a step whose gradients hold inf / nan is skipped and halves the scale, `growth_interval` clean
        # steps in a row double it.  `skipped` counts the skipped steps (they do not advance the optimizer's step count).
        assert precision in (16, 32), precision
        self.precision = precision
        self.graph = bool(graph)
        dev = self.flat.data.device
        if precision == 16:
            # the scaler's state and the optimizer's step count live in device memory: the step never waits for the host.
            # {scale, growth_factor, backoff_factor, growth_interval, clean steps in a row, skipped steps, -, -}
            self._scaler = torch.tensor([init_scale, growth_factor, backoff_factor, growth_interval, 0, 0, 0, 0], dtype=torch.float32, device=dev)
            self._absmax = torch.zeros(1, dtype=torch.float32, device=dev)
        self._graphs = {}
        self._one = torch.ones(1, dtype=torch.float32, device=dev)
        self._build_pack_list()
        if self.graph or precision == 16:
            self._step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            self._lr_dev = torch.full((8,), lr, dtype=torch.float32, device=dev)   # ESMI_TRAIN_ADAMW_HYPER_FLOATS: [0] = lr

    # precision 16: read-outs of the device-side state (each is one small device -> host copy; the step itself never reads them)
    @property
    def scale(self):
        return float(self._scaler[0].item()) if self.precision == 16 else 1.0

    @property
    def skipped(self):
        return int(self._scaler[5].item()) if self.precision == 16 else 0

    @property
    def growth_interval(self):
        return int(self._scaler[3].item())

    @growth_interval.setter
    def growth_interval(self, n):
        self._scaler[3] = float(n)

    @property
    def t(self):
        return int(self._step_dev.item()) if self.precision == 16 else self._t

    @t.setter
    def t(self, v):
        self._t = int(v)
        if self.precision == 16 and hasattr(self, "_step_dev"):
            self._step_dev.fill_(int(v))

    # ---- checkpoint / resume (what Lightning's ModelCheckpoint keeps: weights, hyper-parameters, the optimizer's state_dict)
    def hyper_parameters(self):
        """The constructor arguments of the reference's LightningModule (model.py:104-121) that describe this network: what
        `load_from_checkpoint` needs next to the weights."""
        pe, dec = self.net.encoder, self.net.decoder
        e = pe.encoder
        return {"depth": e.depth, "reduction": e.embed_dim // pe.dim, "head": e.heads[0], "embed_dim": e.embed_dim,
                "kernel_size": e.kernels[0], "expansion": e.expansion, "n_blocks": dec.n_blocks, "block_depth": dec.block_depth,
                "decoder_kernel_size": dec.kernel_size, "lr": self.lr, "weight_decay": self.wd}

    def optimizer_state_dict(self):
        """`torch.optim.AdamW(net.parameters()).state_dict()` as the reference's optimizer would hold it after `self.t` steps:
        per-parameter `step` / `exp_avg` / `exp_avg_sq` (views of the flat moment buffers, cloned), indexed by the parameter's
        position in `net.parameters()`; parameters that never received a gradient (the bins, the two unreached LayerNorms) have
        no state, as in torch."""
        f = self.flat
        index = {id(p): j for j, p in enumerate(self.net.parameters())}
        state, off = {}, 0
        t_now = self.t                          # (precision 16: a device read -- once, not per parameter)
        for p in f.params:
            k = p.numel()
            if t_now > 0:
                state[index[id(p)]] = {"step": torch.tensor(float(t_now)), "exp_avg": f.m[off:off + k].view(p.shape).clone(),
                                       "exp_avg_sq": f.v[off:off + k].view(p.shape).clone()}
            off += (k + 3) & ~3
        group = {"lr": self.lr, "initial_lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": True, "params": list(range(len(index)))}
        return {"state": state, "param_groups": [group]}

    def state_dict(self):
        """A Lightning-shaped checkpoint dict: `state_dict` ('phoneme2mel.<key>' weights), `hyper_parameters` (what
        `EfficientSpeech.load_from_checkpoint` / the reference's constructor need), `optimizer_states` = [a torch AdamW
        state_dict] (loads into `torch.optim.AdamW(net.parameters())`), `global_step`.  The reference's own
        `load_from_checkpoint` additionally expects `hifigan.*` weights (its module owns the vocoder, model.py:148) and a
        `preprocess_config` argument: pass `strict=False, preprocess_config=...` there, or attach the vocoder's weights under
        `hifigan.` before saving."""
        ck = {"state_dict": {"phoneme2mel." + k: v.detach().clone() for k, v in self.net.state_dict().items()},
              "hyper_parameters": self.hyper_parameters(), "optimizer_states": [self.optimizer_state_dict()],
              "global_step": self.t, "epoch": int(getattr(self, "epoch", 0))}
        if self.precision == 16:                   # torch.amp.GradScaler.state_dict()'s keys (Lightning stores it next to the optimizer)
            sc = self._scaler.cpu().tolist()
            ck["scaler"] = {"scale": sc[0], "growth_factor": sc[1], "backoff_factor": sc[2], "growth_interval": int(sc[3]),
                            "_growth_tracker": int(sc[4])}
        return ck

    def load_state_dict(self, ckpt):
        """Resume: weights into the flat buffer's views (in place), the AdamW moments and step count from the optimizer state
        (torch's AdamW state_dict layout, as `state_dict()` writes it).  Only optimizer states written by `TrainStep.state_dict()` --
        i.e. over `net.parameters()` of the acoustic model alone -- load: the reference's own optimizer also owns the vocoder's
        parameters (model.py:148, 279-283), so its state has a different parameter list and is refused below.  `ckpt['epoch']`
        (when present) is returned so that the caller can pass it to `fit(first_epoch=...)`."""
        sd = {k[len("phoneme2mel."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("phoneme2mel.")}
        own = dict(self.net.state_dict())
        missing = [k for k in own if k not in sd]
        if missing:
            raise RuntimeError(f"checkpoint lacks {missing[:3]} ...")
        with torch.no_grad():
            for k, t in own.items():
                t.copy_(sd[k])                      # in place: parameters stay views of the flat buffer
        o = ckpt["optimizer_states"][0]
        f = self.flat
        index = {id(p): j for j, p in enumerate(self.net.parameters())}
        if len(o["param_groups"][0]["params"]) != len(index):
            raise RuntimeError("optimizer state was saved for a different parameter list")
        f.m.zero_()
        f.v.zero_()
        steps, off = set(), 0
        for p in f.params:
            k = p.numel()
            st = o["state"].get(index[id(p)])
            if st is not None:
                f.m[off:off + k].copy_(st["exp_avg"].reshape(-1))
                f.v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(st["step"]))
            off += (k + 3) & ~3
        if len(steps) > 1:
            raise RuntimeError(f"per-parameter step counts differ: {sorted(steps)}")
        self.t = steps.pop() if steps else 0
        g = o["param_groups"][0]
        # the BASE learning rate: under a scheduler torch stores it as `initial_lr` and keeps the scheduled value in `lr` (0 at the
        # first step of the reference's LambdaLR warm-up) -- `fit` derives every step's rate from the base one
        self.lr, self.wd, self.betas, self.eps = g.get("initial_lr", g["lr"]), g["weight_decay"], tuple(g["betas"]), g["eps"]
        if self.graph and self.precision != 16:
            self._step_dev.fill_(self.t)
        if self.precision == 16 and "scaler" in ckpt:
            sc = ckpt["scaler"]
            self._scaler[:5] = torch.tensor([sc["scale"], sc["growth_factor"], sc["backoff_factor"], sc["growth_interval"],
                                             sc["_growth_tracker"]], dtype=torch.float32)
        self._invalidate_packed()
        self.epoch = int(ckpt.get("epoch", 0))
        return self.epoch

    def _invalidate_packed(self):
        """The optimizer kernel wrote the weights behind torch's version counters: drop EVERY packed copy the inference path
        keeps (each module's `_PackCache` attributes, whatever their names -- the decoder has two -- and the one-call
        forward's argument block) so that the next eval forward re-packs from the updated parameters."""
        caches = getattr(self, "_pack_caches", None)
        if caches is None or self._pack_caches_n != sum(1 for _ in self.net.modules()):
            # collected once (and again only if the module tree changed): this runs every step, graph replays included, and walking
            # every module's __dict__ was pure host time in a loop that is host-dispatch bound
            mods = list(self.net.modules())
            self._pack_caches_n = len(mods)
            self._pack_owners = [m for m in mods if hasattr(m, "_launch")]      # the one-call forward keeps an argument block (lazily created)
            caches = self._pack_caches = [v for m in mods for v in m.__dict__.values() if isinstance(v, networks._PackCache)]
        for v in caches:
            v.invalidate()
        for m in self._pack_owners:
            m._fwd_ident = None

    def _build_pack_list(self):
        """Persistent GEMM copies (forward layout, data-gradient layout) of every dense convolution / Linear weight, re-made by ONE
        launch per step (esmi_train_pack_weights_f32) instead of one pack launch per operator call (40 per step)."""
        mods = [m for m in self.net.modules() if isinstance(m, (torch.nn.Conv1d, torch.nn.ConvTranspose1d, torch.nn.Linear))
                and getattr(m, "groups", 1) == 1 and getattr(m.weight, "_esmi_grad_view", None) is not None]
        self._pack_n = len(mods)
        self._pack_descs = (_lib.ConvDesc * max(1, len(mods)))()
        self._pack_w = (C.c_void_p * max(1, len(mods)))()
        lib = _lib.load()
        for j, m in enumerate(mods):
            w = m.weight
            tr = isinstance(m, torch.nn.ConvTranspose1d)
            w3 = w if w.dim() == 3 else w.unsqueeze(-1)
            c_in, c_out, k = (w3.shape[0], w3.shape[1], w3.shape[2]) if tr else (w3.shape[1], w3.shape[0], w3.shape[2])
            stride, pad = (m.stride[0], m.padding[0]) if w.dim() == 3 else (1, 0)
            d = _lib.ConvDesc(1, 1, c_in, 1, c_out, k, stride, pad, 1, 1 if tr else 0, 0)
            nb = lib.esmi_train_conv_workspace_bytes(C.byref(d))
            w._esmi_packed = (torch.zeros(nb, dtype=torch.uint8, device=w.device), torch.zeros(nb, dtype=torch.uint8, device=w.device))
            d.packed_fwd, d.packed_grad = _ptr(w._esmi_packed[0]), _ptr(w._esmi_packed[1])
            self._pack_descs[j] = d
            self._pack_w[j] = w.data_ptr()

    def _fwd_bwd(self, x, y):
        """Forward, loss, backward, and the one launch that finishes every parameter gradient: everything up to the exchange.
        Capturable as a hipGraph (no host reads, no collectives)."""
        global _DIRECT_GRADS, PRECISION, _REDUCE_Q, _PACKED_VALID, _LOSS_SEED
        f = self.flat
        f.zero_grad()
        rq = (_lib.ReduceQueue(), [])
        amp = self.precision == 16
        old_precision, PRECISION = PRECISION, self.precision
        lib0, st0 = _rt(f.data)
        try:
            if USE_MATRIX_PIPE and self._pack_n:   # every operator of this step reads these copies of the current weights
                lib0.esmi_train_pack_weights_f32(self._pack_descs, self._pack_w, self._pack_n, st0)
                _PACKED_VALID = True
            _LOSS_SEED = self._scaler[:1] if amp else False
            parts, total = training_loss(self.net, x, y)
            _DIRECT_GRADS = True                   # one backward on a zeroed buffer: operators write parameter gradients in place
            _REDUCE_Q = rq
            if amp:
                total.backward(gradient=self._scaler[0].reshape(total.shape))     # GradScaler.scale(loss).backward(): the seed is the device-side scale
            else:
                total.backward(gradient=self._one.reshape(total.shape))
        finally:
            _DIRECT_GRADS = False
            _REDUCE_Q = None
            _PACKED_VALID = False
            _LOSS_SEED = None
            PRECISION = old_precision
        lib0.esmi_train_reduce_flush_f32(C.byref(rq[0]), st0)      # every queued parameter-gradient reduction in one launch
        rq[1].clear()
        return loss_vector(parts, total)

    def _optimize(self, lr, on_device):
        """The exchange (one all-reduce of the flat gradient buffer when data-parallel) and the optimizer launch.  `on_device`: step
        count / learning rate (/ the loss scaler's state) are read from device memory (graph replay, precision 16)."""
        f = self.flat
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(f.grad, op=dist.ReduceOp.SUM, group=self.group)
            f.grad.div_(self.world)                # DDP averages (train.py:66-70 runs Lightning's default DDP strategy)
        lib, st = _rt(f.data)
        if self.precision == 16:
            # GradScaler.step + .update on the device: max|g| of the (still scaled) gradients -- absmax's integer max orders inf and
            # nan above every finite magnitude -- decides inside the optimizer launch whether the update runs (on g / scale) or is
            # skipped with the scale backed off; nothing is read back.
            lib.esmi_absmax_f32(_ptr(f.grad), f.grad.numel(), _ptr(self._absmax), st)
            lib.esmi_train_adamw_graph_f32(_ptr(f.data), _ptr(f.grad), _ptr(f.m), _ptr(f.v), f.data.numel(), _ptr(self._lr_dev),
                                           self.betas[0], self.betas[1], self.eps, self.wd, _ptr(self._step_dev), _ptr(self._absmax),
                                           _ptr(self._scaler), st)
        elif on_device:
            lib.esmi_train_adamw_graph_f32(_ptr(f.data), _ptr(f.grad), _ptr(f.m), _ptr(f.v), f.data.numel(), _ptr(self._lr_dev),
                                           self.betas[0], self.betas[1], self.eps, self.wd, _ptr(self._step_dev), None, None, st)
        else:
            lib.esmi_train_adamw_f32(_ptr(f.data), _ptr(f.grad), _ptr(f.m), _ptr(f.v), f.data.numel(), lr, self.betas[0],
                                     self.betas[1], self.eps, self.wd, self.t, 1.0, st)

    def _body(self, x, y, lr, graph):
        losses = self._fwd_bwd(x, y)
        self._optimize(lr, graph)
        return losses

    def step(self, x, y, lr=None):
        """One training step.  graph=True: a hipGraph per batch SHAPE replays the step (inputs are copied into its static buffers) --
        the whole step on one GPU; data-parallel, everything up to the gradient all-reduce (the collective and the optimizer launch
        follow eagerly)."""
        if self.precision != 16:
            self._t += 1                           # (precision 16 counts on the device: a skipped step does not advance it)
        lr = self.lr if lr is None else lr
        on_device = self.graph or self.precision == 16
        if on_device:
            self._lr_dev[:1].fill_(lr)
        if not self.graph:
            out = self._body(x, y, lr, False)
            self._invalidate_packed()
            return out
        whole = self.world == 1                    # the collective stays outside a graph
        captured = self._body if whole else (lambda sx, sy, lr_, g_: self._fwd_bwd(sx, sy))
        key = tuple((k, tuple(v.shape)) for k, v in sorted({**x, **{"y." + k: v for k, v in y.items()}}.items()) if torch.is_tensor(v))
        ent = self._graphs.get(key)
        if ent is None:
            sx = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in x.items()}
            sy = {k: v.clone() for k, v in y.items()}
            side = torch.cuda.Stream(device=self.flat.data.device)          # one eager step on a side stream warms the allocator
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                out = self._body(sx, sy, lr, True)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static_out = captured(sx, sy, lr, True)
            self._graphs[key] = (g, sx, sy, static_out)
            self._invalidate_packed()
            return out.clone()                     # (the capture itself does not execute: this batch's step ran eagerly above)
        g, sx, sy, static_out = ent
        for k, v in x.items():
            if torch.is_tensor(v):
                sx[k].copy_(v)
        for k, v in y.items():
            sy[k].copy_(v)
        g.replay()
        if not whole:
            self._optimize(lr, True)
        self._invalidate_packed()
        return static_out.clone()


def lr_at_epoch(epoch, base_lr=1e-3, warmup=50, total=5000, min_lr=0.0):
    """model.py:77-101 + :281: LambdaLR(linear warm-up over `warmup` epochs, then cosine decay to `total`), stepped once per epoch
    by Lightning's default scheduler interval -- epoch 0 therefore trains at lr 0, as in the reference."""
    import math
    if epoch < warmup:
        lam = float(epoch) / float(max(1, warmup))
    else:
        lam = max(min_lr, 0.5 * (1.0 + math.cos(math.pi * float(epoch - warmup) / float(max(1, total - warmup)))))
    return base_lr * lam


def to_device(batch, device):
    """Move the tensors of a datamodule batch (x or y dict; x also carries the raw `text` list) to `device`."""
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}


def fit(step, loader, epochs, device, base_lr=None, warmup=50, total=5000, first_epoch=0, log=None):
    """The reference's training loop (train.py:66-76 -> Lightning `trainer.fit`) reduced to what it computes: for every epoch,
    the scheduled learning rate; for every batch of the datamodule, one `TrainStep.step`; at the end of the epoch the means of the
    five losses over the epoch's steps, averaged over the data-parallel ranks (model.py:227-242: `on_train_epoch_end` logs them
    with `sync_dist=True`) -- one 5-float all-reduce per epoch on `step.group`.  Returns the per-epoch records."""
    base_lr = step.lr if base_lr is None else base_lr
    history = []
    for epoch in range(first_epoch, first_epoch + epochs):
        lr = lr_at_epoch(epoch, base_lr, warmup, total)
        acc, n = None, 0
        for x, y in loader:
            losses = step.step(to_device(x, device), to_device(y, device), lr=lr)
            acc = losses.clone() if acc is None else acc + losses        # five scalars per step: bookkeeping, like self.log()
            n += 1
        mean = acc / max(n, 1) if acc is not None else None
        if mean is not None and step.world > 1:
            import torch.distributed as dist
            dist.all_reduce(mean, op=dist.ReduceOp.SUM, group=step.group)
            mean = mean / step.world
        history.append({"epoch": epoch, "lr": lr, "losses": mean.cpu().tolist() if mean is not None else None})
        step.epoch = epoch + 1          # what `state_dict()` records: the epoch a resumed `fit(first_epoch=...)` starts with
        if log:
            log(history[-1])
    return history


def synthetic_batch(B, T, dur, device, seed=5):
    """A seeded teacher-forced batch of the shapes datamodule.collate_fn produces (datamodule.py:29-76): no padding, D-const."""
    import numpy as np
    from .synth import synth_phonemes
    g = np.random.default_rng(seed)
    ids, mask = synth_phonemes(B, T, seed)
    L = T * dur
    x = {"phoneme": torch.from_numpy(ids), "phoneme_mask": torch.from_numpy(mask),
         "pitch": torch.from_numpy(g.uniform(-3, 10, (B, T)).astype(np.float32)),
         "energy": torch.from_numpy(g.uniform(-2, 8, (B, T)).astype(np.float32)),
         "duration": torch.full((B, T), dur, dtype=torch.int32), "mel_len": torch.full((B,), L, dtype=torch.int32),
         "mel_mask": torch.zeros((B, L), dtype=torch.bool)}
    y = {"mel": torch.from_numpy(g.normal(-5, 2, (B, L, 80)).astype(np.float32))}
    return {k: v.to(device) for k, v in x.items()}, {k: v.to(device) for k, v in y.items()}
