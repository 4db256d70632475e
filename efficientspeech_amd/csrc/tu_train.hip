// esmi C-ABI, translation unit "tu_train.hip": the training step's operators (train_ops.h)
// One of several translation units of libesmi.so (compiled in parallel by __graft_entry__.build(); the simulator build
// tools/wavesim/build.sh compiles the same files with the host compiler).  Internal launchers are declared in launch.h.
#include "launch.h"
#include "train_ops.h"

using namespace esmi;
ESMI_TU_RANGE_SETTER(train)

extern "C" {

// ------------------------------------------------------------------ training step (csrc/train_ops.h)
namespace {
int conv_desc_ok(const esmi_conv_desc* d, ConvDesc* o) {
    if (d && d->precision != 0 && d->precision != 16 && d->precision != 32) return ESMI_ERR_ARG;
    if (!d || d->B <= 0 || d->n_in <= 0 || d->n_out <= 0 || d->c_in <= 0 || d->c_out <= 0 || d->k <= 0 || d->stride <= 0 || d->pad < 0 ||
        d->groups <= 0)
        return ESMI_ERR_ARG;
    if (d->groups != 1 && (d->transposed || d->c_in % d->groups || d->c_out % d->groups)) return ESMI_ERR_UNSUPPORTED;
    *o = ConvDesc{d->B, d->n_in, d->c_in, d->n_out, d->c_out, d->k, d->stride, d->pad, d->groups, d->transposed ? 1 : 0};
    return ESMI_OK;
}
}  // namespace

namespace {
int reduce_flush(esmi_reduce_queue* q, hipStream_t st) {
    if (!q || q->count <= 0) return ESMI_OK;
    if (q->count > ESMI_REDUCE_QUEUE_ITEMS) return ESMI_ERR_ARG;
    ReduceBatch b;
    b.count = q->count;
    long blocks = 0;
    for (int i = 0; i < q->count; ++i) {
        const esmi_reduce_item& s = q->items[i];
        b.items[i] = ReduceItem{s.partial, (long)s.n, (long)s.stride, (long)s.chunks, s.out, (long)s.n0, s.out1};
        b.first[i] = (int)blocks;
        blocks += ((long)s.n + 63) / 64;
        if (blocks > 0x7FFFFFFFL) return ESMI_ERR_ARG;
    }
    b.first[q->count] = (int)blocks;
    if (blocks > 0)
        ESMI_LAUNCH(train_reduce_batch_kernel, dim3((unsigned)blocks), dim3(64 * kReduceGroups), 64 * kReduceGroups * sizeof(float), st, b);
    q->count = 0;
    return launch_status();
}
// second stage of a chunked reduction: now (one launch), or queued for the step's single flush
int reduce_or_defer(esmi_reduce_queue* q, const float* part, long n, long stride, long chunks, float* out, long n0, float* out1, hipStream_t st) {
    if (!q) {
        ESMI_LAUNCH(train_reduce_chunks_kernel, grid1d(n, 64), dim3(64 * kReduceGroups), 64 * kReduceGroups * sizeof(float), st, part, n, stride, chunks,
                    out, n0, out1);
        return launch_status();
    }
    if (q->count >= ESMI_REDUCE_QUEUE_ITEMS)
        if (int rc = reduce_flush(q, st)) return rc;
    q->items[q->count++] = esmi_reduce_item{part, n, stride, chunks, out, n0, out1};
    return ESMI_OK;
}
inline bool wgrad_depthwise(const ConvDesc& c) { return !c.transposed && c.groups == c.c_in && c.c_in == c.c_out && c.k <= 8; }
inline bool wgrad_on_mfma(const ConvDesc& c);
inline unsigned wgrad_tiles(const ConvDesc& c) { return (unsigned)(((c.c_out + 127) / 128) * ((c.c_in + 31) / 32) * c.k); }
inline int wgrad_chunk(const ConvDesc& c) {   // rows per partial sum: fewer for small weights, whose parallelism must come from the chunks
    if (wgrad_on_mfma(c)) {
        // rows per wave, a multiple of 16: as many as give the chip at most two 4-wave workgroups per CU, i.e. ONE round of workgroups
        // (round 5; before: 96 rows whatever the size -- 800 workgroups = 1.6 rounds for a decoder convolution at B = 128, 34 workgroups
        // of 6 trips each for the encoder-side ones)
        const long rows = (long)c.B * c.n_out, want = (rows * wgrad_tiles(c) + 2048 * 16 - 1) / (2048 * 16);
        return 16 * (int)(want < 1 ? 1 : want > 64 ? 64 : want);
    }
    if (wgrad_depthwise(c)) return kTrainChunkDw;
    const long nw = (long)(c.transposed ? c.c_in * c.c_out : c.c_out * (c.c_in / c.groups)) * c.k;
    return nw < 1024 ? 32 : kTrainChunk;
}
inline bool wgrad_on_mfma(const ConvDesc& c) {
    // (train_ops.h train_conv_wgrad_mfma_kernel: 32-bit byte offsets into both tensors, one utterance boundary per 16-row trip at most)
    const long xb = (long)c.B * c.n_in * c.c_in * 4, yb = (long)c.B * c.n_out * c.c_out * 4;
    return c.groups == 1 && c.c_in >= 8 && c.c_out >= 8 && (c.c_out & 3) == 0 && c.n_out >= 16 && c.n_out < (1 << 23) && xb < 0x7FFFFFFFL && yb < 0x7FFFFFFFL;
}
// the matrix-pipe weight gradient (train_ops.h): gridDim = (weight tiles, row groups rounded up to a multiple of 8: XCD-aware order)
int launch_wgrad_mfma(const ConvDesc& c, const float* x, const float* dy, float* part, float* pb, long chunks, long ps, int* amax, hipStream_t st) {
    const dim3 grid(wgrad_tiles(c), (unsigned)(((chunks + 3) / 4 + 7) & ~7L));
    if (c.transposed) {
        static AttrOnce once;
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(train_conv_wgrad_mfma_kernel<true>), once)) return rc;
        ESMI_LAUNCH(train_conv_wgrad_mfma_kernel<true>, grid, dim3(256), kWgradLdsBytes, st, c, x, dy, part, pb, chunks, ps, wgrad_chunk(c), amax);
    } else {
        static AttrOnce once;
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(train_conv_wgrad_mfma_kernel<false>), once)) return rc;
        ESMI_LAUNCH(train_conv_wgrad_mfma_kernel<false>, grid, dim3(256), kWgradLdsBytes, st, c, x, dy, part, pb, chunks, ps, wgrad_chunk(c), amax);
    }
    return launch_status();
}
}
// Dense convolutions (groups == 1) of the training step run on the matrix pipe through the inference path's implicit GEMM
// (convgemm.h) when the caller gives scratch for the tap-major copy of the weight: the forward as it is, the data gradient as
// the transposed problem -- d(Conv1d) is a ConvTranspose1d of dy with the same (Cout, Cin, k) tensor read as (Cin', Cout', k),
// d(ConvTranspose1d) is a Conv1d of dy with (Cin, Cout, k) read as (Cout', Cin', k).  Everything else (depthwise, channel
// counts the GEMM does not take, no scratch) runs the one-thread-per-element kernels of train_ops.h.
size_t esmi_train_conv_workspace_bytes(const esmi_conv_desc* d) {
    ConvDesc c;
    if (conv_desc_ok(d, &c) || c.groups != 1) return 0;
    return align256((size_t)c.k * c.c_out * c.c_in * sizeof(float)) + 256;   // + the data gradient's absmax / scale slots
}
namespace {
// how the GEMM of one direction reads the weight tensor: (supported, needs a copy, read as ConvTranspose1d, taps flipped)
struct GemmWeight { bool ok, copy; int as_convT; bool flipped; int cin, cout; };
GemmWeight gemm_weight(const ConvDesc& c, bool grad) {
    GemmWeight g;
    g.cin = grad ? c.c_out : c.c_in; g.cout = grad ? c.c_in : c.c_out;      // of the GEMM problem
    g.ok = c.groups == 1 && !(g.cin & 7) && !(g.cout == 1 && (grad != (c.transposed != 0) || c.stride != 1));   // the one-channel kernel is a plain conv
    // tap-major (k, cout, cin) of the problem: forward conv / grad of convT read the tensor as Conv1d, the other two as ConvTranspose1d
    g.as_convT = (grad != (c.transposed != 0)) ? 1 : 0;
    // a stride-1 transposed convolution is a plain convolution with the taps reversed and padding k - 1 - pad: packed that way it
    // takes the LDS-staged GEMM kernel (MODE_CONV only) like the forward problem does
    g.flipped = g.as_convT && c.stride == 1 && c.k - 1 - c.pad >= 0;
    g.copy = !(c.k == 1 && !g.as_convT && !grad);   // a Linear's (Cout, Cin) IS its tap-major form
    return g;
}
// one of the two implicit-GEMM problems of a dense conv: returns ESMI_ERR_UNSUPPORTED when the GEMM does not take the shape
// (have_absmax: max|in| already sits in the workspace's absmax slot -- the weight-gradient pass of esmi_train_conv_bwd_f32 left it)
// prepacked: `wt` already holds this step's copy (esmi_train_pack_weights_f32, which also cleared the absmax slot)
struct ConvLnArgs { const float *res, *g, *b; const unsigned char* rowmask; int relu_out; float *pre, *mean, *rstd; };
int train_conv_gemm(const ConvDesc& c, bool grad, const float* in, const float* w, const float* bias, float* out, float* wt,
                    bool amp, hipStream_t st, bool have_absmax = false, bool pack_only = false, bool prepacked = false, int act = 0,
                    const ConvLnArgs* ln = nullptr) {
    const GemmWeight g = gemm_weight(c, grad);
    if (!g.ok || !wt) return ESMI_ERR_UNSUPPORTED;
    // (the fused conv + LayerNorm epilogue exists for 32 / 64 / 128 output channels: refuse BEFORE anything is enqueued -- the caller's
    // two-launch fallback then packs once, not twice)
    if (ln && g.cout != 32 && g.cout != 64 && g.cout != 128) return ESMI_ERR_UNSUPPORTED;
    const int cin = g.cin, cout = g.cout, as_convT = g.as_convT;
    const bool as_flipped_conv = g.flipped;
    const long n = (long)c.k * c.c_out * c.c_in;
    int* amax = reinterpret_cast<int*>(reinterpret_cast<char*>(wt) + align256((size_t)n * sizeof(float)));
    const float* wuse = wt;
    if (!g.copy) {
        wuse = w;                                   // no copy
    } else if (!have_absmax && !prepacked) {   // (with have_absmax the pack ran in the pack_only call that preceded the weight-gradient pass)
        ESMI_LAUNCH(pack_conv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, wt, cout, cin, c.k, as_convT, grad ? amax : nullptr,
                    as_flipped_conv ? 1 : 0);
        if (int rc = launch_status()) return rc;
    }
    if (pack_only) return ESMI_OK;
    ConvGemmP p = conv_defaults();
    if (grad && have_absmax) {
        p.io_scale = reinterpret_cast<const float*>(amax);
    } else if (grad) {   // max|dy| on the device; the GEMM kernels derive the power-of-two scales from it (convgemm.h conv_pow2_scales)
        const long len = (long)c.B * c.n_out * c.c_out, blocks = (len + 256L * 8 - 1) / (256L * 8);
        ESMI_LAUNCH(absmax_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, st, in, len, amax);
        if (int rc = launch_status()) return rc;
        p.io_scale = reinterpret_cast<const float*>(amax);
    }
    p.mode = (as_convT && !as_flipped_conv) ? MODE_CONVT : MODE_CONV;
    p.k = c.k; p.stride = c.stride; p.pad = as_flipped_conv ? c.k - 1 - c.pad : c.pad;
    p.B = c.B; p.n_in = grad ? c.n_out : c.n_in; p.n_out = grad ? c.n_in : c.n_out; p.c_in = cin; p.c_out = cout;
    p.A = in; p.lda = cin; p.W = wuse; p.bias = bias; p.out = out; p.ldo = cout;
    p.amp = amp ? 1 : 0;
    p.act = act;
    p.pw_ok = 1;
    if (ln) {   // act(conv + bias) + res -> LayerNorm (+ ReLU, + row mask) in the GEMM's epilogue; the pre-norm tensor, mean, rstd kept
        p.res = ln->res; p.ldr = cout; p.r_coff = 0;
        p.ln_g = ln->g; p.ln_b = ln->b; p.rowmask = ln->rowmask; p.post_relu = ln->relu_out;
        p.ln_pre = ln->pre; p.ln_mean = ln->mean; p.ln_rstd = ln->rstd;
    }
    return launch_convgemm(p, st);
}
}  // namespace

int esmi_train_pack_weights_f32(const esmi_conv_desc* descs, const float* const* weights, int n, esmi_stream_t stream) {
    if (n < 0 || (n > 0 && (!descs || !weights))) return ESMI_ERR_ARG;
    PackBatch b;
    b.count = 0;
    long nmax = 0;
    auto flush = [&]() -> int {
        if (!b.count) return ESMI_OK;
        const long blocks = (nmax + 255) / 256;
        ESMI_LAUNCH(train_pack_batch_kernel, dim3((unsigned)(blocks > 256 ? 256 : blocks), (unsigned)b.count), dim3(256), 0, S(stream), b);
        b.count = 0; nmax = 0;
        return launch_status();
    };
    for (int i = 0; i < n; ++i) {
        ConvDesc c;
        esmi_conv_desc d = descs[i];
        d.B = d.n_in = d.n_out = 1;                 // (only the weight's shape matters here)
        if (int rc = conv_desc_ok(&d, &c)) return rc;
        if (!weights[i]) return ESMI_ERR_ARG;
        const long nw = (long)c.k * c.c_out * c.c_in;
        for (int grad = 0; grad < 2; ++grad) {
            float* dst = grad ? d.packed_grad : d.packed_fwd;
            if (!dst) continue;
            const GemmWeight g = gemm_weight(c, grad != 0);
            if (!g.ok || !g.copy) continue;         // the direction does not run as a GEMM / reads the tensor as it is
            int* amax = reinterpret_cast<int*>(reinterpret_cast<char*>(dst) + align256((size_t)nw * sizeof(float)));
            b.items[b.count++] = PackItem{weights[i], dst, grad ? amax : nullptr, g.cout, g.cin, c.k, g.as_convT, g.flipped ? 1 : 0};
            nmax = nw > nmax ? nw : nmax;
            if (b.count == kPackBatch) if (int rc = flush()) return rc;
        }
    }
    return flush();
}
int esmi_train_conv_ln_fwd_f32(const esmi_conv_desc* d, const float* x, const float* w, const float* bias, const float* res,
                               const float* ln_g, const float* ln_b, const uint8_t* rowmask, int relu_out, float* y_pre, float* y,
                               float* mean, float* rstd, void* workspace, size_t workspace_bytes, esmi_stream_t stream) {
    ConvDesc c;
    if (int rc = conv_desc_ok(d, &c)) return rc;
    if (!x || !w || !ln_g || !ln_b || !y_pre || !y || !mean || !rstd || d->act < 0 || d->act > ACT_TANH) return ESMI_ERR_ARG;
    if (c.stride != 1 || c.transposed || c.n_in != c.n_out) return ESMI_ERR_UNSUPPORTED;
    if (!(d->packed_fwd || (workspace && workspace_bytes >= esmi_train_conv_workspace_bytes(d)))) return ESMI_ERR_UNSUPPORTED;
    float* wt = d->packed_fwd ? d->packed_fwd : static_cast<float*>(workspace);
    const ConvLnArgs ln{res, ln_g, ln_b, rowmask, relu_out ? 1 : 0, y_pre, mean, rstd};
    return train_conv_gemm(c, false, x, w, bias, y, wt, d->precision == 16, S(stream), false, false, d->packed_fwd != nullptr, d->act, &ln);
}
int esmi_train_conv_fwd_f32(const esmi_conv_desc* d, const float* x, const float* w, const float* bias, float* y, void* workspace,
                            size_t workspace_bytes, esmi_stream_t stream) {
    ConvDesc c;
    if (int rc = conv_desc_ok(d, &c)) return rc;
    if (!x || !w || !y || d->act < 0 || d->act > ACT_TANH) return ESMI_ERR_ARG;
    if (d->packed_fwd || (workspace && workspace_bytes >= esmi_train_conv_workspace_bytes(d))) {
        float* wt = d->packed_fwd ? d->packed_fwd : static_cast<float*>(workspace);
        const int rc = train_conv_gemm(c, false, x, w, bias, y, wt, d->precision == 16, S(stream), false, false, d->packed_fwd != nullptr, d->act);
        if (rc != ESMI_ERR_UNSUPPORTED) return rc;
    }
    const long n = (long)c.B * c.n_out * c.c_out;
    if (wgrad_depthwise(c) && c.stride == 1 && (c.c_out & 3) == 0) {
        ESMI_LAUNCH(train_conv_dw_kernel, grid1d((long)c.B * ((c.n_out + kDwRows - 1) / kDwRows) * (c.c_out / 4)), dim3(256), 0, S(stream), c, x, w, bias, y, 0);
    } else {
        ESMI_LAUNCH(train_conv_fwd_kernel, grid1d(n), dim3(256), 0, S(stream), c, x, w, bias, y);
    }
    if (int rc = launch_status()) return rc;
    if (d->act) {   // the plain kernels have no epilogue: the activation as its own launch, in place
        ESMI_LAUNCH(train_act_fwd_kernel, grid1d(n), dim3(256), 0, S(stream), y, n, d->act, y);
        return launch_status();
    }
    return ESMI_OK;
}
int esmi_train_conv_dgrad_f32(const esmi_conv_desc* d, const float* dy, const float* w, float* dx, void* workspace,
                              size_t workspace_bytes, esmi_stream_t stream) {
    ConvDesc c;
    if (int rc = conv_desc_ok(d, &c)) return rc;
    if (!dy || !w || !dx) return ESMI_ERR_ARG;
    if (d->packed_grad || (workspace && workspace_bytes >= esmi_train_conv_workspace_bytes(d))) {
        float* wt = d->packed_grad ? d->packed_grad : static_cast<float*>(workspace);
        const int rc = train_conv_gemm(c, true, dy, w, nullptr, dx, wt, d->precision == 16, S(stream), false, false, d->packed_grad != nullptr);
        if (rc != ESMI_ERR_UNSUPPORTED) return rc;
    }
    const long n = (long)c.B * c.n_in * c.c_in;
    if (wgrad_depthwise(c) && c.stride == 1 && (c.c_out & 3) == 0) {
        ESMI_LAUNCH(train_conv_dw_kernel, grid1d((long)c.B * ((c.n_in + kDwRows - 1) / kDwRows) * (c.c_in / 4)), dim3(256), 0, S(stream), c, dy, w, nullptr, dx, 1);
        return launch_status();
    }
    ESMI_LAUNCH(train_conv_dgrad_kernel, grid1d(n), dim3(256), 0, S(stream), c, dy, w, dx);
    return launch_status();
}
size_t esmi_train_conv_wgrad_workspace_bytes(const esmi_conv_desc* d) {
    ConvDesc c;
    if (conv_desc_ok(d, &c)) return 0;
    const long nw = (long)(c.transposed ? c.c_in * c.c_out : c.c_out * (c.c_in / c.groups)) * c.k;
    const long chunks = train_chunks((long)c.B * c.n_out, wgrad_chunk(c));
    return (size_t)chunks * (size_t)(nw + c.c_out) * sizeof(float);
}
static int conv_wgrad_impl(const esmi_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias, void* workspace,
                           size_t workspace_bytes, esmi_reduce_queue* defer, esmi_stream_t stream) {
    ConvDesc c;
    if (int rc = conv_desc_ok(d, &c)) return rc;
    if (!x || !dy || !dw || !workspace) return ESMI_ERR_ARG;
    if (workspace_bytes < esmi_train_conv_wgrad_workspace_bytes(d)) return ESMI_ERR_WORKSPACE;
    const long nw = (long)(c.transposed ? c.c_in * c.c_out : c.c_out * (c.c_in / c.groups)) * c.k;
    const bool mfma = wgrad_on_mfma(c), depthwise = wgrad_depthwise(c);
    const long rows = (long)c.B * c.n_out, chunks = train_chunks(rows, wgrad_chunk(c));
    float* part = static_cast<float*>(workspace);   // [chunk][weight partials (nw) | bias partials (c_out)]
    float* pb = part + nw;
    const long ps = nw + c.c_out;
    if (mfma) {   // dense: one wave per (128 output channels x 32 input channels, tap, chunk) on the fp32 MFMA; bias partials from the tap 0, ci0 = 0 waves
        if (int rc = launch_wgrad_mfma(c, x, dy, part, dbias ? pb : nullptr, chunks, ps, nullptr, S(stream))) return rc;
    } else if (depthwise && (c.c_out & 3) == 0 && c.stride == 1 && c.n_in == c.n_out) {
        ESMI_LAUNCH(train_conv_wgrad_dw4_kernel, dim3((unsigned)((c.c_out + 127) / 128), (unsigned)chunks), dim3(256), kDwSub * 9 * 128 * sizeof(float),
                    S(stream), c, x, dy, part, dbias ? pb : nullptr, ps);
    } else if (depthwise) {
        ESMI_LAUNCH(train_conv_wgrad_dw_kernel, dim3(grid1d(c.c_out, 64), (unsigned)chunks), dim3(64), 0, S(stream), c, x, dy, part,
                    dbias ? pb : nullptr, ps);
    } else {
        ESMI_LAUNCH(train_conv_wgrad_kernel, dim3(grid1d(nw, 64), (unsigned)chunks), dim3(64), 0, S(stream), c, x, dy, part, ps, wgrad_chunk(c));
        if (int rc = launch_status()) return rc;
        if (dbias) ESMI_LAUNCH(train_colsum_kernel, dim3(grid1d(c.c_out, 64), (unsigned)chunks), dim3(64), 0, S(stream), dy, rows, c.c_out, pb, ps, wgrad_chunk(c));
    }
    if (int rc = launch_status()) return rc;
    // weight and bias partials in ONE reduction: elements >= nw of a partial row are the bias sums
    // (the matrix-pipe kernel already summed its four waves: one partial row per workgroup)
    return reduce_or_defer(defer, part, dbias ? ps : nw, ps, mfma ? (chunks + 3) / 4 : chunks, dw, nw, dbias, S(stream));
}
int esmi_train_conv_wgrad_f32(const esmi_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias, void* workspace,
                              size_t workspace_bytes, esmi_stream_t stream) {
    return conv_wgrad_impl(d, x, dy, dw, dbias, workspace, workspace_bytes, nullptr, stream);
}
int esmi_train_reduce_flush_f32(esmi_reduce_queue* q, esmi_stream_t stream) {
    if (!q) return ESMI_ERR_ARG;
    return reduce_flush(q, S(stream));
}
size_t esmi_train_conv_bwd_workspace_bytes(const esmi_conv_desc* d) {
    return align256(esmi_train_conv_wgrad_workspace_bytes(d)) + esmi_train_conv_workspace_bytes(d);
}
int esmi_train_conv_bwd_f32(const esmi_conv_desc* d, const float* x, const float* dy, const float* w, float* dx, float* dw, float* dbias,
                            void* workspace, size_t workspace_bytes, esmi_reduce_queue* defer, esmi_stream_t stream) {
    ConvDesc c;
    if (int rc = conv_desc_ok(d, &c)) return rc;
    if (!x || !dy || !w || !dx || !dw || !workspace) return ESMI_ERR_ARG;
    if (workspace_bytes < esmi_train_conv_bwd_workspace_bytes(d)) return ESMI_ERR_WORKSPACE;
    const size_t wg_bytes = align256(esmi_train_conv_wgrad_workspace_bytes(d)), gemm_bytes = esmi_train_conv_workspace_bytes(d);
    char* ws = static_cast<char*>(workspace);
    const bool pre = d->packed_grad != nullptr;
    float* wt = pre ? d->packed_grad : reinterpret_cast<float*>(ws + wg_bytes);
    // dense shapes whose two gradients both run on the matrix pipe: tap-major weight copy (zeroes the absmax slot) -> weight gradient
    // (leaves max|dy| in the slot) -> data-gradient GEMM scaled by it.  Everything else: the two stand-alone entry points.
    const bool fused = gemm_bytes > 0 && wgrad_on_mfma(c) &&
                       train_conv_gemm(c, true, dy, w, nullptr, dx, wt, d->precision == 16, S(stream), false, true, pre) == ESMI_OK;
    if (!fused) {
        if (c.c_out == 1 && c.k == 1 && c.stride == 1 && c.pad == 0 && c.groups == 1 && !c.transposed && c.n_in == c.n_out && wgrad_chunk(c) <= 64) {
            // a Linear down to one channel: the three gradients in one launch (train_ops.h train_lin1_bwd_kernel)
            const long rows = (long)c.B * c.n_out, chunks = train_chunks(rows, wgrad_chunk(c));
            float* part = reinterpret_cast<float*>(ws);
            const long ps = c.c_in + 1;
            const bool al16 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(w)) & 15) == 0;
            float* pbias = dbias ? part + c.c_in : nullptr;
#define ESMI_LIN1(LPR) ESMI_LAUNCH(train_lin1_bwd4_kernel<LPR>, dim3((unsigned)chunks), dim3(64), 0, S(stream), x, dy, w, rows, dx, part, pbias, ps)
            if (al16 && wgrad_chunk(c) == 32 && c.c_in == 16) ESMI_LIN1(4);
            else if (al16 && wgrad_chunk(c) == 32 && c.c_in == 32) ESMI_LIN1(8);
            else if (al16 && wgrad_chunk(c) == 32 && c.c_in == 64) ESMI_LIN1(16);
            else if (al16 && wgrad_chunk(c) == 32 && c.c_in == 128) ESMI_LIN1(32);
            else if (al16 && wgrad_chunk(c) == 32 && c.c_in == 256) ESMI_LIN1(64);
            else
                ESMI_LAUNCH(train_lin1_bwd_kernel, dim3((unsigned)chunks), dim3(64), 0, S(stream), x, dy, w, rows, c.c_in, dx, part, pbias, ps, wgrad_chunk(c));
#undef ESMI_LIN1
            if (int rc = launch_status()) return rc;
            return reduce_or_defer(defer, part, dbias ? ps : (long)c.c_in, ps, chunks, dw, c.c_in, dbias, S(stream));
        }
        if (int rc = esmi_train_conv_dgrad_f32(d, dy, w, dx, gemm_bytes ? wt : nullptr, gemm_bytes, stream)) return rc;
        return conv_wgrad_impl(d, x, dy, dw, dbias, workspace, wg_bytes, defer, stream);
    }
    const long nw = (long)(c.transposed ? c.c_in * c.c_out : c.c_out * (c.c_in / c.groups)) * c.k;
    int* amax = reinterpret_cast<int*>(reinterpret_cast<char*>(wt) + align256((size_t)nw * sizeof(float)));
    const long rows = (long)c.B * c.n_out, chunks = train_chunks(rows, wgrad_chunk(c));
    float* part = reinterpret_cast<float*>(ws);
    float* pb = part + nw;
    const long ps = nw + c.c_out;
    if (int rc = launch_wgrad_mfma(c, x, dy, part, dbias ? pb : nullptr, chunks, ps, amax, S(stream))) return rc;
    if (int rc = reduce_or_defer(defer, part, dbias ? ps : nw, ps, (chunks + 3) / 4, dw, nw, dbias, S(stream))) return rc;
    return train_conv_gemm(c, true, dy, w, nullptr, dx, wt, d->precision == 16, S(stream), true, false, pre);
}
int esmi_train_layernorm_fwd_f32(const float* x, const float* g, const float* b, int64_t rows, int C, float* y, float* mean,
                                 float* rstd, const float* res, float* xsum, const uint8_t* rowmask, int relu_out, esmi_stream_t stream) {
    if (!x || !g || !b || !y || !mean || !rstd || rows <= 0 || C <= 0 || (res && !xsum)) return ESMI_ERR_ARG;
    uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(b);
    if (res) al |= reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(xsum);
    if ((al & 15) == 0 && (C == 32 || C == 64 || C == 128 || C == 256)) {   // rows in registers, 16-byte accesses
#define ESMI_LN4(LPR) ESMI_LAUNCH(train_ln_fwd4_kernel<LPR>, grid1d(rows, 4 * (64 / LPR)), dim3(256), 0, S(stream), x, g, b, (long)rows, 1e-5f, y, mean, rstd, res, xsum, rowmask, relu_out ? 1 : 0)
        if (C == 32) ESMI_LN4(8); else if (C == 64) ESMI_LN4(16); else if (C == 128) ESMI_LN4(32); else ESMI_LN4(64);
#undef ESMI_LN4
        return launch_status();
    }
    ESMI_LAUNCH(train_ln_fwd_kernel, grid1d(rows, 4), dim3(256), 0, S(stream), x, g, b, (long)rows, C, 1e-5f, y, mean, rstd, res, xsum, rowmask, relu_out ? 1 : 0);
    return launch_status();
}
size_t esmi_train_layernorm_bwd_workspace_bytes(int64_t rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    return (size_t)train_chunks(rows, C <= 256 ? kLnRows : kTrainChunk) * 2 * C * sizeof(float);
}
int esmi_train_layernorm_bwd_f32(const float* x, const float* g, const float* mean, const float* rstd, const float* dy,
                                 int64_t rows, int C, float* dx, float* dg, float* db, void* workspace, size_t workspace_bytes,
                                 esmi_reduce_queue* defer, const uint8_t* rowmask, int in_act, const float* y_relu, esmi_stream_t stream) {
    if (!x || !g || !mean || !rstd || !dy || !dx || !dg || !db || !workspace || rows <= 0 || C <= 0) return ESMI_ERR_ARG;
    if (in_act != 0 && in_act != ACT_RELU && in_act != ACT_TANH) return ESMI_ERR_ARG;
    if (workspace_bytes < esmi_train_layernorm_bwd_workspace_bytes(rows, C)) return ESMI_ERR_WORKSPACE;
    float* part = static_cast<float*>(workspace);
    long chunks;
    if (C <= 256) {   // dx and the parameter partials in one pass
        chunks = train_chunks(rows, kLnRows);
        uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(g);
        if (y_relu) al |= reinterpret_cast<uintptr_t>(y_relu);
        if ((al & 15) == 0 && (C == 32 || C == 64 || C == 128)) {   // rows in registers, 16-byte accesses, all rows of a wave loaded up front
#define ESMI_LNB4(LPR) ESMI_LAUNCH(train_ln_bwd4_kernel<LPR>, dim3((unsigned)chunks), dim3(256), 4 * 2 * 256 * sizeof(float), S(stream), x, g, mean, rstd, dy, (long)rows, dx, part, rowmask, in_act, y_relu)
            if (C == 32) ESMI_LNB4(8); else if (C == 64) ESMI_LNB4(16); else ESMI_LNB4(32);
#undef ESMI_LNB4
            if (int rc = launch_status()) return rc;
            return reduce_or_defer(defer, part, 2L * C, 2L * C, chunks, dg, (long)C, db, S(stream));
        }
        ESMI_LAUNCH(train_ln_bwd_fused_kernel, dim3((unsigned)chunks), dim3(256), 4 * 2 * 256 * sizeof(float), S(stream), x, g, mean, rstd, dy, (long)rows, C, dx, part, rowmask, in_act, y_relu);
        if (int rc = launch_status()) return rc;
    } else {
        chunks = train_chunks(rows);
        ESMI_LAUNCH(train_ln_bwd_dx_kernel, grid1d(rows, 4), dim3(256), 0, S(stream), x, g, mean, rstd, dy, (long)rows, C, dx, rowmask, in_act, y_relu);
        if (int rc = launch_status()) return rc;
        ESMI_LAUNCH(train_ln_bwd_params_kernel, dim3(grid1d(C, 64), (unsigned)chunks), dim3(64), 0, S(stream), x, mean, rstd, dy, (long)rows, C, part, rowmask, y_relu);
        if (int rc = launch_status()) return rc;
    }
    return reduce_or_defer(defer, part, 2L * C, 2L * C, chunks, dg, (long)C, db, S(stream));   // partial rows are [dg (C) | db (C)]: two outputs
}
int esmi_train_act_fwd_f32(const float* x, int64_t n, int kind, float* y, esmi_stream_t stream) {
    if (!x || !y || n <= 0 || kind < ACT_RELU || kind > ACT_TANH) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_act_fwd_kernel, grid1d(n), dim3(256), 0, S(stream), x, (long)n, kind, y);
    return launch_status();
}
int esmi_train_act_bwd_f32(const float* saved, const float* dy, int64_t n, int kind, float* dx, esmi_stream_t stream) {
    if (!saved || !dy || !dx || n <= 0 || kind < ACT_RELU || kind > ACT_TANH) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_act_bwd_kernel, grid1d(n), dim3(256), 0, S(stream), saved, dy, (long)n, kind, dx);
    return launch_status();
}
namespace {
// row segments per (utterance, head) of the LDS-staged attention kernels: enough workgroups for ~4 per CU, at least 4 rows per wave
inline unsigned attn_row_segments(int heads_total, int N) {
    int rs = (1024 + heads_total - 1) / heads_total;
    const int max_rs = (N + 15) / 16;
    if (rs > max_rs) rs = max_rs;
    return (unsigned)(rs < 1 ? 1 : rs);
}
}  // namespace
int esmi_train_attention_fwd_f32(const float* qkv, int B, int N, int C, int h, float* P, float* ctx, esmi_stream_t stream) {
    if (!qkv || !P || !ctx || B <= 0 || N <= 0 || C <= 0 || h <= 0 || C % h) return ESMI_ERR_ARG;
    const size_t lds = train_attn_lds_bytes(N, C);
    if (lds <= 150 * 1024) {   // K and V of a head staged in LDS
        static AttrOnce once;
        if (lds > 48 * 1024)
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(train_attn_fwd_lds_kernel), once)) return rc;
        ESMI_LAUNCH(train_attn_fwd_lds_kernel, dim3((unsigned)(B * h), attn_row_segments(B * h, N)), dim3(256), lds, S(stream), qkv, B, N, C, h,
                    1.0f / sqrtf((float)(C / h)), P, ctx);
        return launch_status();
    }
    ESMI_LAUNCH(train_attn_fwd_kernel, dim3((unsigned)((long)B * h * N)), dim3(64), 0, S(stream), qkv, B, N, C, h, 1.0f / sqrtf((float)(C / h)), P, ctx);
    return launch_status();
}
int esmi_train_attention_bwd_f32(const float* qkv, const float* P, const float* dctx, int B, int N, int C, int h, float* dS,
                                 float* dqkv, esmi_stream_t stream) {
    if (!qkv || !P || !dctx || !dS || !dqkv || B <= 0 || N <= 0 || C <= 0 || h <= 0 || C % h) return ESMI_ERR_ARG;
    const float scale = 1.0f / sqrtf((float)(C / h));
    const size_t lds = train_attn_lds_bytes(N, C);
    if (lds <= 150 * 1024) {
        static AttrOnce once;
        if (lds > 48 * 1024)
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(train_attn_bwd_rows_lds_kernel), once)) return rc;
        ESMI_LAUNCH(train_attn_bwd_rows_lds_kernel, dim3((unsigned)(B * h), attn_row_segments(B * h, N)), dim3(256), lds, S(stream), qkv, P, dctx,
                    B, N, C, h, scale, dS, dqkv);
    } else {
        ESMI_LAUNCH(train_attn_bwd_rows_kernel, dim3((unsigned)((long)B * h * N)), dim3(64), 0, S(stream), qkv, P, dctx, B, N, C, h, scale, dS, dqkv);
    }
    if (int rc = launch_status()) return rc;
    ESMI_LAUNCH(train_attn_bwd_cols_kernel, grid1d((long)B * h * N * C), dim3(256), 0, S(stream), qkv, P, dS, dctx, B, N, C, h, scale, dqkv);
    return launch_status();
}
int esmi_train_embedding_fwd_f32(const int32_t* ids, const float* table, int64_t rows, int V, int C, float* out, esmi_stream_t stream) {
    if (!ids || !table || !out || rows <= 0 || V <= 0 || C <= 0) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_embed_fwd_kernel, grid1d(rows * C), dim3(256), 0, S(stream), ids, table, (long)rows, V, C, out);
    return launch_status();
}
size_t esmi_train_embedding_bwd_workspace_bytes(int64_t rows, int V, int C) {
    return rows > 0 && V > 0 && C > 0 ? (size_t)train_chunks(rows) * V * C * sizeof(float) : 0;
}
int esmi_train_embedding_bwd_f32(const int32_t* ids, const float* dy, int64_t rows, int V, int C, int padding_idx, float* dtable,
                                 void* workspace, size_t workspace_bytes, esmi_reduce_queue* defer, esmi_stream_t stream) {
    if (!ids || !dy || !dtable || !workspace || rows <= 0 || V <= 0 || C <= 0) return ESMI_ERR_ARG;
    if (workspace_bytes < esmi_train_embedding_bwd_workspace_bytes(rows, V, C)) return ESMI_ERR_WORKSPACE;
    const long chunks = train_chunks(rows), n = (long)V * C;
    float* part = static_cast<float*>(workspace);
    ESMI_LAUNCH(train_embed_bwd_kernel, dim3(grid1d(n, 64), (unsigned)chunks), dim3(64), 0, S(stream), ids, dy, (long)rows, V, C, padding_idx, part);
    if (int rc = launch_status()) return rc;
    return reduce_or_defer(defer, part, n, n, chunks, dtable, -1, nullptr, S(stream));
}
int esmi_train_mask_rows_f32(const float* x, const uint8_t* mask, int64_t rows, int C, float* y, esmi_stream_t stream) {
    if (!x || !mask || !y || rows <= 0 || C <= 0) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_mask_rows_kernel, grid1d(rows * C), dim3(256), 0, S(stream), x, mask, (long)rows, C, y);
    return launch_status();
}
int esmi_train_add_f32(const float* a, const float* b, int64_t n, float* y, esmi_stream_t stream) {
    if (!a || !b || !y || n <= 0) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_add_kernel, grid1d(n), dim3(256), 0, S(stream), a, b, (long)n, y);
    return launch_status();
}
int esmi_train_copy_cols_f32(const float* src, int ld_src, int col_src, float* dst, int ld_dst, int col_dst, int64_t rows, int C,
                             esmi_stream_t stream) {
    if (!src || !dst || rows <= 0 || C <= 0 || col_src < 0 || col_dst < 0 || col_src + C > ld_src || col_dst + C > ld_dst) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_copy_cols_kernel, grid1d(rows * C), dim3(256), 0, S(stream), src, ld_src, col_src, dst, ld_dst, col_dst, (long)rows, C);
    return launch_status();
}
int esmi_train_cat_f32(float* const* parts, const int* widths, int n, int64_t rows, float* cat, const uint8_t* rowmask,
                       unsigned masked_parts, int backward, esmi_stream_t stream) {
    if (!parts || !widths || n <= 0 || n > 8 || rows <= 0 || !cat || (masked_parts && !rowmask)) return ESMI_ERR_ARG;
    CatArgs a;
    memset(&a, 0, sizeof a);
    int col = 0;
    for (int j = 0; j < n; ++j) {
        if (!parts[j] || widths[j] <= 0) return ESMI_ERR_ARG;
        a.part[j] = parts[j]; a.width[j] = widths[j]; a.col[j] = col;
        col += widths[j];
    }
    a.n = n; a.tot = col; a.rows = (long)rows; a.cat = cat; a.rowmask = rowmask; a.masked = masked_parts; a.backward = backward ? 1 : 0;
    ESMI_LAUNCH(train_cat_kernel, grid1d((long)rows * col), dim3(256), 0, S(stream), a);
    return launch_status();
}
int esmi_train_repeat_fwd_f32(const float* feat, const int32_t* cum, int B, int T, int C, int L, float* out, esmi_stream_t stream) {
    if (!feat || !cum || !out || B <= 0 || T <= 0 || C <= 0 || L <= 0) return ESMI_ERR_ARG;
    if ((C & 3) == 0 && (long)B * L * (C / 4) < 0x7FFFFFFFL && ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        ESMI_LAUNCH(train_repeat_fwd4_kernel, grid1d((long)B * L * (C / 4)), dim3(256), 0, S(stream), feat, cum, B, T, C / 4, L, out);
        return launch_status();
    }
    ESMI_LAUNCH(train_repeat_fwd_kernel, grid1d((long)B * L * C), dim3(256), 0, S(stream), feat, cum, B, T, C, L, out);
    return launch_status();
}
int esmi_train_repeat_bwd_f32(const float* dout, const int32_t* cum, int B, int T, int C, int L, float* dfeat, esmi_stream_t stream) {
    if (!dout || !cum || !dfeat || B <= 0 || T <= 0 || C <= 0 || L <= 0) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_repeat_bwd_kernel, grid1d((long)B * T * C), dim3(256), 0, S(stream), dout, cum, B, T, C, L, dfeat);
    return launch_status();
}
int esmi_train_loss_f32(const esmi_train_loss_args* a, esmi_stream_t stream) {
    if (!a || !a->mel_pred || !a->mel || !a->pitch_pred || !a->pitch || !a->energy_pred || !a->energy || !a->dur_pred || !a->dur ||
        !a->out || !a->d_mel || !a->d_pitch || !a->d_energy || !a->d_dur || !a->scratch || a->B <= 0 || a->T <= 0 || a->L <= 0 || a->n_mel <= 0)
        return ESMI_ERR_ARG;
    static_assert(ESMI_TRAIN_LOSS_SCRATCH_FLOATS >= kLossBlocks * 6, "scratch size in the header");
    LossP p = {a->mel_pred, a->mel, a->pitch_pred, a->pitch, a->energy_pred, a->energy, a->dur_pred, a->dur, a->mel_mask, a->ph_mask,
               a->B, a->T, a->L, a->n_mel, a->out, a->d_mel, a->d_pitch, a->d_energy, a->d_dur, a->scratch, a->grad_seed};
    ESMI_LAUNCH(train_loss_partial_kernel, dim3(kLossBlocks), dim3(256), 256 * sizeof(float), S(stream), p);
    if (int rc = launch_status()) return rc;
    ESMI_LAUNCH(train_loss_final_kernel, dim3(1), dim3(256), 256 * sizeof(float), S(stream), p);
    if (int rc = launch_status()) return rc;
    const long nm = (long)a->B * a->L * a->n_mel, np_ = (long)a->B * a->T;
    ESMI_LAUNCH(train_loss_grad_kernel, grid1d(nm > np_ ? nm : np_), dim3(256), 0, S(stream), p);
    return launch_status();
}
int esmi_train_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                         double weight_decay, int step, double grad_scale, esmi_stream_t stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1 || !(grad_scale > 0.0)) return ESMI_ERR_ARG;
    // 1 - beta^step in double (-expm1(step * log beta)): in fp32, 1 - 0.999f^t carries ~6e-5 relative error at small t
    const double bc1 = -expm1((double)step * log(beta1)), bc2 = -expm1((double)step * log(beta2));
    AdamWScalars h = {(float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)(1.0 - lr * weight_decay),
                      (float)(lr / bc1), (float)sqrt(bc2), (float)grad_scale};
    ESMI_LAUNCH(train_adamw_kernel, grid1d(n), dim3(256), 0, S(stream), p, g, m, v, (long)n, h);
    return launch_status();
}

int esmi_train_adamw_graph_f32(float* p, const float* g, float* m, float* v, int64_t n, float* hyper_dev, double beta1, double beta2,
                               double eps, double weight_decay, int32_t* step_dev, const float* grad_absmax, float* scaler_state,
                               esmi_stream_t stream) {
    if (!p || !g || !m || !v || !hyper_dev || !step_dev || n <= 0 || ((scaler_state != nullptr) != (grad_absmax != nullptr))) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_bump_step_kernel, dim3(1), dim3(64), 0, S(stream), step_dev, hyper_dev, beta1, beta2, weight_decay, grad_absmax,
                scaler_state);
    if (int rc = launch_status()) return rc;
    AdamWScalars h = {(float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, 0.0f, 0.0f, 0.0f, 1.0f};
    ESMI_LAUNCH(train_adamw_dev_kernel, grid1d(n), dim3(256), 0, S(stream), p, g, m, v, (long)n, h, (const float*)hyper_dev);
    return launch_status();
}

}  // extern "C"
