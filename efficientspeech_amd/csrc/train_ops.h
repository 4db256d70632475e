// Training-step operators (model.py:167-226, 279-283): forward ops that keep what backward reads, their data-
// and weight-gradient kernels, the fused masked loss and AdamW.
//
// FIRST CORRECT VERSION: every kernel is one thread per OUTPUT element with plain fp32 FMA loops -- no atomics anywhere, so
// a training step is bitwise reproducible -- on channels-last activations (B, n, C) and weights in CHECKPOINT layout (what
// the optimizer updates): Conv1d (Cout, Cin/groups, k), ConvTranspose1d (Cin, Cout, k), Linear (Cout, Cin) = Conv1d with k = 1.
// The matrix-pipe versions of the same contractions (convgemm.h for the forward and data gradients, a split-K MFMA
// reduction for the weight gradients) replace these one by one behind the same entry points.
#pragma once
#include "esmi_dev.h"

namespace esmi {

struct ConvDesc {   // mirrors esmi_conv_desc (include/esmi.h), without its trailing `precision` (a property of the GEMM launch)
    int B, n_in, c_in, n_out, c_out, k, stride, pad, groups, transposed;
};

// weight element for (output channel co, input channel ci, tap j); false when the pair is not connected (grouped conv)
__device__ __forceinline__ bool conv_w_index(const ConvDesc& d, int co, int ci, int j, long* idx) {
    if (d.transposed) { *idx = ((long)ci * d.c_out + co) * d.k + j; return true; }        // (Cin, Cout, k), groups == 1
    const int cig = d.c_in / d.groups, cog = d.c_out / d.groups;
    const int g = co / cog;
    if (ci / cig != g) return false;
    *idx = ((long)co * cig + (ci - g * cig)) * d.k + j;                                     // (Cout, Cin/groups, k)
    return true;
}
// input position feeding output position t through tap j (or -1)
__device__ __forceinline__ int conv_in_pos(const ConvDesc& d, int t, int j) {
    if (!d.transposed) {
        const int ti = t * d.stride + j - d.pad;
        return (ti >= 0 && ti < d.n_in) ? ti : -1;
    }
    const int q = t + d.pad - j;                      // ConvTranspose1d: out[n*stride + j - pad] += in[n] * W[:, :, j]
    if (q < 0 || q % d.stride) return -1;
    const int ti = q / d.stride;
    return ti < d.n_in ? ti : -1;
}

// y[b, t, co] = bias[co] + sum_j sum_ci x[b, in_pos(t, j), ci] * w(co, ci, j)
static __global__ void train_conv_fwd_kernel(const ConvDesc d, const float* __restrict__ x, const float* __restrict__ w,
                                      const float* __restrict__ bias, float* __restrict__ y) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long)d.B * d.n_out * d.c_out) return;
    const int co = (int)(q % d.c_out), t = (int)((q / d.c_out) % d.n_out), b = (int)(q / ((long)d.c_out * d.n_out));
    const int cig = d.transposed ? d.c_in : d.c_in / d.groups, ci0 = d.transposed ? 0 : (co / (d.c_out / d.groups)) * cig;
    float acc = bias ? bias[co] : 0.0f;
    for (int j = 0; j < d.k; ++j) {
        const int ti = conv_in_pos(d, t, j);
        if (ti < 0) continue;
        const float* xr = x + ((long)b * d.n_in + ti) * d.c_in;
        for (int c = 0; c < cig; ++c) {
            long wi;
            conv_w_index(d, co, ci0 + c, j, &wi);
            acc = fmaf(xr[ci0 + c], w[wi], acc);
        }
    }
    y[q] = acc;
}

// depthwise, stride 1 (MelDecoder's k = 5 convs): one thread per (kDwRows consecutive rows, 4 channels), 16-byte accesses -- the rows
// slide through registers (kDwRows + k - 1 loads for kDwRows outputs instead of k each), the k x 4 weights are read once.  `grad`
// runs the data gradient: dx[t] = sum_j dy[t + pad - j] w[j] = the same correlation with the taps reversed and padding k - 1 - pad.
// Requires k <= 8; weights (C, 1, k)
constexpr int kDwRows = 8;
static __global__ __launch_bounds__(256) void train_conv_dw_kernel(const ConvDesc d, const float* __restrict__ in, const float* __restrict__ w,
                                     const float* __restrict__ bias, float* __restrict__ out, int grad) {
    const int c4 = d.c_out >> 2;
    const int n_o = grad ? d.n_in : d.n_out, n_i = grad ? d.n_out : d.n_in;
    const int tb = (n_o + kDwRows - 1) / kDwRows;
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long)d.B * tb * c4) return;
    const int c = (int)(q % c4) * 4, t0 = (int)((q / c4) % tb) * kDwRows, b = (int)(q / ((long)c4 * tb));
    const int pad = grad ? d.k - 1 - d.pad : d.pad;
    f32x4 wt[8];                                       // wt[j][e]: tap j of channel c + e, in correlation order
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) wt[j][e] = j < d.k ? w[(long)(c + e) * d.k + (grad ? d.k - 1 - j : j)] : 0.0f;
    }
    const f32x4 b4 = (bias && !grad) ? ld4(bias + c) : zero4();
    f32x4 acc[kDwRows];
#pragma unroll
    for (int o = 0; o < kDwRows; ++o) acc[o] = b4;
    const float* base = in + (long)b * n_i * d.c_out + c;
    // every input row first, unconditionally (clamped into the utterance, zeroed where outside: round 5 -- a load under a per-row
    // condition is waited for before the next one is issued), then the products: row t0 - pad + r feeds output o through tap j = r - o
    f32x4 v[kDwRows + 7];
#pragma unroll
    for (int r = 0; r < kDwRows + 7; ++r) {
        const int ti = t0 - pad + r, tc = ti < 0 ? 0 : (ti >= n_i ? n_i - 1 : ti);
        v[r] = r < kDwRows + d.k - 1 ? ld4(base + (long)tc * d.c_out) : zero4();
        if (ti != tc) v[r] = zero4();
    }
#pragma unroll
    for (int r = 0; r < kDwRows + 7; ++r) {
#pragma unroll
        for (int o = 0; o < kDwRows; ++o) {
            const int j = r - o;
            if (j >= 0 && j < 8) {                     // (compile-time after unrolling; taps >= k hold zero weights)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[o][e] = fmaf(v[r][e], wt[j][e], acc[o][e]);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < kDwRows; ++o)
        if (t0 + o < n_o) *reinterpret_cast<f32x4*>(out + ((long)b * n_o + t0 + o) * d.c_out + c) = acc[o];
}

// dx[b, ti, ci] = sum over (t, j) with in_pos(t, j) == ti, and co connected to ci, of dy[b, t, co] * w(co, ci, j)
static __global__ void train_conv_dgrad_kernel(const ConvDesc d, const float* __restrict__ dy, const float* __restrict__ w,
                                        float* __restrict__ dx) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long)d.B * d.n_in * d.c_in) return;
    const int ci = (int)(q % d.c_in), ti = (int)((q / d.c_in) % d.n_in), b = (int)(q / ((long)d.c_in * d.n_in));
    const int cog = d.transposed ? d.c_out : d.c_out / d.groups, co0 = d.transposed ? 0 : (ci / (d.c_in / d.groups)) * cog;
    float acc = 0.0f;
    for (int j = 0; j < d.k; ++j) {
        int t;
        if (!d.transposed) {                          // ti = t*stride + j - pad
            const int r = ti + d.pad - j;
            if (r < 0 || r % d.stride) continue;
            t = r / d.stride;
        } else {                                      // t = ti*stride + j - pad
            t = ti * d.stride + j - d.pad;
        }
        if (t < 0 || t >= d.n_out) continue;
        const float* dr = dy + ((long)b * d.n_out + t) * d.c_out;
        for (int c = 0; c < cog; ++c) {
            long wi;
            conv_w_index(d, co0 + c, ci, j, &wi);
            acc = fmaf(dr[co0 + c], w[wi], acc);
        }
    }
    dx[q] = acc;
}

// Reductions over the rows (B * n positions) run in two deterministic stages: stage 1 gives every (output element, chunk of
// kTrainChunk rows) its own thread and writes a partial sum, stage 2 adds the partials of an element in chunk order.
constexpr int kTrainChunk = 256;
constexpr int kTrainChunkDw = 128;     // rows per workgroup of the depthwise weight gradient (8 sub-chunks of 16 rows, summed in LDS; 256: 300 workgroups at B = 128, 38 us)
// (rows per wave of the matrix-pipe weight gradient: tu_train.hip wgrad_chunk -- a multiple of 16 sized for one round of workgroups.
// Round 3, when a trip was a chain of exposed load latencies, a fixed 48 ... 192 rows made no difference: 5.2-5.7 ms per B = 128 step.)
__host__ __device__ inline long train_chunks(long rows, int chunk = kTrainChunk) { return (rows + chunk - 1) / chunk; }

// dw(co, ci, j) = sum over (b, t) of dy[b, t, co] * x[b, in_pos(t, j), ci]: partial[chunk][weight element in checkpoint order]
static __global__ void train_conv_wgrad_kernel(const ConvDesc d, const float* __restrict__ x, const float* __restrict__ dy,
                                        float* __restrict__ partial, long pstride, int chunk) {
    const int cig = d.transposed ? d.c_out : d.c_in / d.groups;      // middle extent of the checkpoint layout
    const int outer = d.transposed ? d.c_in : d.c_out;
    const long nw = (long)outer * cig * d.k;
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nw) return;
    const int j = (int)(q % d.k), mid = (int)((q / d.k) % cig), out = (int)(q / ((long)d.k * cig));
    int co, ci;
    if (d.transposed) { ci = out; co = mid; }
    else { co = out; ci = (co / (d.c_out / d.groups)) * cig + mid; }
    const long rows = (long)d.B * d.n_out, r0 = (long)blockIdx.y * chunk;
    const long r1 = r0 + chunk < rows ? r0 + chunk : rows;
    float acc = 0.0f;
    for (long r = r0; r < r1; ++r) {
        const int b = (int)(r / d.n_out), t = (int)(r - (long)b * d.n_out);
        const int ti = conv_in_pos(d, t, j);
        if (ti < 0) continue;
        acc = fmaf(dy[r * d.c_out + co], x[((long)b * d.n_in + ti) * d.c_in + ci], acc);
    }
    partial[(long)blockIdx.y * pstride + q] = acc;
}
// A Linear down to ONE channel (the variance predictors' last layer, networks.py:162): its three gradients in one launch -- dx = dy w,
// and per chunk of <= 64 rows the partial sums of dw = sum_r dy[r] x[r, :] and db = sum_r dy[r] (row order / a fixed butterfly:
// reproducible).  One wave per chunk: lane l keeps dy of row l, the lanes then run across the channels (coalesced) with dy handed
// round by v_readlane.  (Before: data gradient, weight gradient and column sum as three launches, the weight gradient with a 64-bit
// division per row and thread: 5 + 20 + 5 us per predictor at B = 128.)
static __global__ __launch_bounds__(64) void train_lin1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ w,
                                                            long rows, int C, float* __restrict__ dx, float* __restrict__ partial,
                                                            float* __restrict__ partial_bias, long pstride, int chunk) {
    const int lane = lane_id();
    const long r0 = (long)blockIdx.x * chunk;
    const int nr = (int)(r0 + chunk < rows ? chunk : rows - r0);
    const float dyv = lane < nr ? dy[r0 + lane] : 0.0f;
    const int dyb = __builtin_bit_cast(int, dyv);
    for (int c0 = 0; c0 < C; c0 += 64) {               // (every lane runs the loop: the hand-round is a wave-level exchange)
        const int c = c0 + lane;
        const bool ok = c < C;
        const float wc = ok ? w[c] : 0.0f;
        float acc = 0.0f;
#pragma unroll 8
        for (int rr = 0; rr < nr; ++rr) {
            const float g = __builtin_bit_cast(float, bcast_i(dyb, rr));
            if (ok) {
                acc = fmaf(g, x[(r0 + rr) * C + c], acc);
                dx[(r0 + rr) * C + c] = g * wc;
            }
        }
        if (ok) partial[(long)blockIdx.x * pstride + c] = acc;
    }
    if (partial_bias) {
        float t = dyv;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) t += shfl_xor_f(t, m);
        if (lane == 0) partial_bias[(long)blockIdx.x * pstride] = t;
    }
}
// The same for C = 4 LPR in {16 .. 256} on 16-byte aligned tensors and a chunk of 32 rows: LPR lanes per row, 64 / LPR rows per pass,
// 16-byte accesses, every load of the chunk in flight before the first use; the lane groups' sums are added by a fixed butterfly.
template <int LPR>
static __global__ __launch_bounds__(64) void train_lin1_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ w,
                                                             long rows, float* __restrict__ dx, float* __restrict__ partial,
                                                             float* __restrict__ partial_bias, long pstride) {
    constexpr int RPW = 64 / LPR, C = 4 * LPR, CH = 32, NP = CH / RPW;
    const int lane = lane_id(), sub = lane / LPR, l = lane % LPR;
    const long r0 = (long)blockIdx.x * CH + sub;                       // this lane's row of pass p: r0 + RPW p
    const f32x4 wv = ld4(w + 4 * l);
    f32x4 xv[NP];
    float g[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const long r = r0 + RPW * p, rc = r < rows ? r : rows - 1;
        xv[p] = ld4(x + rc * C + 4 * l);
        g[p] = r < rows ? dy[rc] : 0.0f;
    }
    f32x4 acc = zero4();
    float gs = 0.0f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const long r = r0 + RPW * p;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] = fmaf(g[p], xv[p][e], acc[e]); o[e] = g[p] * wv[e]; }
        gs += g[p];
        if (r < rows) *reinterpret_cast<f32x4*>(dx + r * C + 4 * l) = o;
    }
#pragma unroll
    for (int k = LPR; k < 64; k <<= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += shfl_xor_f(acc[e], k);
        gs += shfl_xor_f(gs, k);
    }
    if (sub == 0) {                                   // (pstride = C + 1: the partial rows are not 16-byte aligned)
#pragma unroll
        for (int e = 0; e < 4; ++e) partial[(long)blockIdx.x * pstride + 4 * l + e] = acc[e];
    }
    if (partial_bias && lane == 0) partial_bias[(long)blockIdx.x * pstride] = gs;
}
// depthwise (groups == C, one input channel per output channel, k <= 8): lanes across the channels (coalesced), every thread
// keeps the k tap sums and the bias sum of its channel over the chunk's rows
static __global__ void train_conv_wgrad_dw_kernel(const ConvDesc d, const float* __restrict__ x, const float* __restrict__ dy,
                                           float* __restrict__ partial, float* __restrict__ partial_bias, long pstride) {
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (c >= d.c_out) return;
    const long rows = (long)d.B * d.n_out, r0 = (long)blockIdx.y * kTrainChunkDw, r1 = r0 + kTrainChunkDw < rows ? r0 + kTrainChunkDw : rows;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bs = 0.0f;
    for (long r = r0; r < r1; ++r) {
        const int b = (int)(r / d.n_out), t = (int)(r - (long)b * d.n_out);
        const float g = dy[r * d.c_out + c];
        bs += g;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j >= d.k) break;
            const int ti = conv_in_pos(d, t, j);
            if (ti >= 0) acc[j] = fmaf(g, x[((long)b * d.n_in + ti) * d.c_in + c], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (j < d.k) partial[(long)blockIdx.y * pstride + (long)c * d.k + j] = acc[j];
    if (partial_bias) partial_bias[(long)blockIdx.y * pstride + c] = bs;
}

// The same for C % 4 == 0, stride 1, n_in == n_out (every depthwise convolution of the model): a 256-thread workgroup = 32 channel
// quads x kDwSub row sub-chunks of one chunk.  A thread walks its kTrainChunkDw / kDwSub rows in groups of 8: 8 rows of dY and 8 + k - 1 rows of X
// (16-byte loads; the input row of tap j for flat row r is flat row r + j - pad) feed 8 k fused multiply-adds per channel -- 2.5 loads
// per row instead of 6 scalar ones; a group that touches an utterance edge takes the tap-by-tap path.  The sub-chunks' sums are
// added in LDS in sub-chunk order (fixed: reproducible) and one partial row per workgroup goes out.
constexpr int kDwSub = 8;
static __global__ __launch_bounds__(256) void train_conv_wgrad_dw4_kernel(const ConvDesc d, const float* __restrict__ x, const float* __restrict__ dy,
                                                                   float* __restrict__ partial, float* __restrict__ partial_bias, long pstride) {
    ESMI_DYN_LDS(red);   // [kDwSub][9][128]: sub-chunk, tap (8 = bias), channel within the workgroup's 128
    const int lq = (int)threadIdx.x & 31, sub = (int)threadIdx.x >> 5;
    const int c = ((int)blockIdx.x * 32 + lq) * 4;
    const long rows = (long)d.B * d.n_out;
    const long c0 = (long)blockIdx.y * kTrainChunkDw + sub * (kTrainChunkDw / kDwSub);
    const long c1 = c0 + kTrainChunkDw / kDwSub < rows ? c0 + kTrainChunkDw / kDwSub : rows;
    f32x4 acc[8], bs = zero4();
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = zero4();
    if (c < d.c_out) {
        const int n = d.n_out, reach = d.k - 1 - d.pad;
        for (long g0 = c0; g0 < c1; g0 += 8) {
            const int t0 = (int)(g0 % n);
            // every load of the group first, unconditionally (rows clamped into the tensor; what must not count is masked below).  Round 5:
            // a group that touched an utterance edge took a tap-by-tap path of ~40 dependent loads, and one such lane held its whole wave --
            // a tenth of the waves ran ~20x longer than the rest (38 us per launch at B = 128 for 78 MB).
            f32x4 v[15], g[8];
#pragma unroll
            for (int r = 0; r < 15; ++r) {
                long xr = g0 - d.pad + r;
                xr = xr < 0 ? 0 : (xr >= rows ? rows - 1 : xr);
                v[r] = r < 8 + d.k - 1 ? ld4(x + xr * d.c_in + c) : zero4();
            }
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const bool ok = g0 + o < c1;
                g[o] = ld4(dy + (ok ? g0 + o : c1 - 1) * d.c_out + c);
                if (!ok) g[o] = zero4();
                bs += g[o];
            }
            if (g0 + 8 <= c1 && t0 >= d.pad && t0 + 7 + reach < n) {     // the whole group and all its taps inside one utterance
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j >= d.k) break;
#pragma unroll
                    for (int o = 0; o < 8; ++o)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(g[o][e], v[o + j][e], acc[j][e]);
                }
            } else {                                                      // an utterance edge inside the group: the same sums, tap by tap masked
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j >= d.k) break;
#pragma unroll
                    for (int o = 0; o < 8; ++o) {
                        const int ti = (t0 + o) % n + j - d.pad;              // (the row's own utterance: rows past the chunk have g = 0)
                        const bool ok = ti >= 0 && ti < n;
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(g[o][e], ok ? v[o + j][e] : 0.0f, acc[j][e]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(red + ((sub * 9 + j) * 32 + lq) * 4) = acc[j];
    *reinterpret_cast<f32x4*>(red + ((sub * 9 + 8) * 32 + lq) * 4) = bs;
    __syncthreads();
    // thread -> (tap or bias, channel): 9 x 128 sums over the 8 sub-chunks
    for (int q = (int)threadIdx.x; q < 9 * 128; q += 256) {
        const int j = q >> 7, cc = q & 127, ch = (int)blockIdx.x * 128 + cc;
        if (ch >= d.c_out || (j < 8 && j >= d.k)) continue;
        float sum = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < kDwSub; ++s2) sum += red[(s2 * 9 + j) * 128 + cc];
        if (j < 8) partial[(long)blockIdx.y * pstride + (long)ch * d.k + j] = sum;
        else if (partial_bias) partial_bias[(long)blockIdx.y * pstride + ch] = sum;
    }
}

// The same partial sums for DENSE convolutions on the matrix pipe: dW_j = dY^T X_j is a GEMM with the rows as the contraction
// (v_mfma_f32_32x32x2_f32: exact fp32 products, k-ordered accumulation).  One wave = a block of 128 output channels x one
// 32-channel input tile x one tap x one chunk of rows.  Lane (i, kh) reads 16 bytes of dY -- channels 4i..4i+3 of row r + kh,
// so the 32 lanes of a half cover the whole 128-channel block, fully coalesced -- and one float of X; the four values feed four
// MFMAs whose tiles are the channel sets {4m + t}: four MFMAs per two loads, dY read once per input tile instead of once per
// (input tile, output tile).  The (tap 0, first input tile) waves also sum dY's columns: the bias gradient.
constexpr int kWgradWaves = 4;                      // waves (= row chunks) per workgroup, summed in LDS before anything is written
constexpr int kWgradLdsBytes = kWgradWaves * 64 * 64 * 4;   // [wave][accumulator register 0..63][lane]
// One trip's loads of the matrix-pipe weight gradient, ALL issued unconditionally and without a branch (rows past the chunk, taps
// outside the sequence and channels past the tensor read a valid address and are zeroed by the returned mask when they are used --
// straight-line code is what lets the compiler count the loads in flight: with a branch around a load it waits for vmcnt(0) and
// the next trip's loads are no longer in flight under this trip's products): U rows rbase + S u per lane, 16 bytes of dY and one
// float of X each.  One 32-bit division per trip (rows < 2^31 and n_out >= 8 > S u, so a row is at most ONE utterance further than
// the trip's first: tu_train.hip wgrad_on_mfma).  Bit u of the result: dY valid, bit 8 + u: X valid.
template <int U, int S, bool TR>
__device__ __forceinline__ unsigned wgrad_issue(const ConvDesc& d, const float* __restrict__ x, const float* __restrict__ dy, long rbase,
                                                long r1, int j, int co4, bool co_ok, int ci, bool ci_ok, f32x4 (&a)[U], float (&bv)[U]) {
    const unsigned rc = (unsigned)(rbase < r1 ? rbase : r1 - 1), n = (unsigned)d.n_out;
    const unsigned b = rc / n, t = rc - b * n;
    const int colA = co_ok ? co4 : 0, colB = ci_ok ? ci : 0;
    unsigned ok = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool rv = rbase + S * u < r1;
        unsigned tu = t + S * u, bu = b;
        const bool wrap = tu >= n;
        tu = rv ? (wrap ? tu - n : tu) : t;
        bu = rv && wrap ? bu + 1 : bu;
        int ti;
        bool tv;
        if (!TR) {
            ti = (int)tu * d.stride + j - d.pad;
            tv = ti >= 0 && ti < d.n_in;
        } else {                                      // ConvTranspose1d: out[n stride + j - pad] += in[n] W[:, :, j]
            const int q = (int)tu + d.pad - j, qq = q < 0 ? 0 : q;
            ti = qq / d.stride;
            tv = q >= 0 && ti * d.stride == qq && ti < d.n_in;
        }
        ti = tv ? ti : 0;
        a[u] = ld4(dy + (long)(rv ? rc + (unsigned)(S * u) : rc) * d.c_out + colA);
        bv[u] = x[((long)bu * d.n_in + ti) * d.c_in + colB];
        ok |= ((rv && co_ok) ? 1u : 0u) << u | ((rv && ci_ok && tv) ? 1u : 0u) << (8 + u);
    }
    return ok;
}
template <bool TR>
static __global__ __launch_bounds__(256, 2) void train_conv_wgrad_mfma_kernel(const ConvDesc d, const float* __restrict__ x,
                                                                    const float* __restrict__ dy, float* __restrict__ partial,
                                                                    float* __restrict__ partial_bias, long chunks, long pstride,
                                                                    int chunk_rows, int* __restrict__ dy_absmax = nullptr) {
    // dy_absmax (optional): max |dy| as a by-product -- the (tap 0, first input tile) waves read every element of dY exactly once;
    // an integer atomicMax of the magnitude's bit pattern is order-independent (the data-gradient GEMM that follows derives its
    // power-of-two operand scale from it: no separate pass over dY).
    // The four waves of a workgroup take four consecutive row chunks of the SAME weight tile; their accumulators are added in LDS in
    // wave order (fixed order: reproducible) and ONE partial per workgroup leaves the CU -- the partial sums of the two-stage
    // reduction were this kernel's memory traffic (600 chunks x 64 KB for a decoder convolution at B = 128): four times less now.
    //
    // Round 5: (a) a trip's loads are issued one trip AHEAD of its products (the kernel was a chain of exposed load latencies: ~3 us
    // per 16-row trip whatever the size); (b) chunk_rows follows the problem (tu_train.hip wgrad_chunk: small problems get their
    // parallelism from short chunks); (c) the weight tiles that read the SAME rows of dY are neighbours in one XCD's dispatch order
    // (workgroup n runs on XCD n % 8: with the tile as the fast grid index the four tiles of a decoder convolution sat on four XCDs
    // and each pulled dY through its own L2).  gridDim = (tiles, row groups rounded up to a multiple of 8).
    ESMI_DYN_LDS(red);
    const int lane = lane_id(), i = lane & 31, kh = lane >> 5, w = uniform_i(wave_id());   // (a scalar: the chunk's buffer resource lives in SGPRs)
    const int tci = (d.c_in + 31) / 32;
    const long wg = (long)blockIdx.y * gridDim.x + blockIdx.x, slot = wg >> 3;
    const int tile = (int)(slot % gridDim.x);
    const long ygrp = (slot / gridDim.x) * 8 + (wg & 7);
    if (ygrp * kWgradWaves >= chunks) return;          // (the whole workgroup: in front of every barrier)
    const int j = tile % d.k, ci0 = ((tile / d.k) % tci) * 32, cb = (tile / (d.k * tci)) * 128;
    const long chunk = ygrp * kWgradWaves + w;
    const long rows = (long)d.B * d.n_out;
    long r0 = chunk * chunk_rows, r1 = r0 + chunk_rows < rows ? r0 + chunk_rows : rows;
    const int co4 = cb + 4 * i;                         // this lane's four output channels
    const bool vec_ok = co4 + 3 < d.c_out, ci_ok = ci0 + i < d.c_in;
    f32x16 acc[4] = {zero16(), zero16(), zero16(), zero16()};
    f32x4 bsum = zero4();
    float amax_f = 0.0f;
#if ESMI_CHAIN_SPLIT
    // fp32-accurate split products on the bf16 matrix pipe (esmi_dev.h split_bf16x3: three 8-bit pieces per value = all 24 significand
    // bits, six products, bf16's exponent range -- gradients as small as 1e-30 lose nothing, which binary16 pieces would need a scale
    // for): the contraction runs over the ROWS, so a lane's eight k-slots of a 16-row step are eight consecutive rows -- lane (i, kh)
    // loads dY channels 4i .. 4i+3 and X channel ci0 + i of rows r + 8 kh + (0..7); each of the four dY channels is the A operand of
    // one of four MFMA tiles (channel sets {4m + t}).  24 v_mfma_f32_32x32x16_bf16 (768 cycles) per 16 rows instead of 32
    // v_mfma_f32_32x32x2_f32 (2048 cycles).
    constexpr int kU = 8, kTrip = 16;
#else
    constexpr int kU = 4, kTrip = 8;                   // rows r + 2 u + kh: 8 rows per trip, 16 v_mfma_f32_32x32x2_f32
#endif
    constexpr int kS = kTrip == 16 ? 1 : 2;
    const int lane_row = kTrip == 16 ? 8 * kh : kh;
    const bool stats = j == 0 && ci0 == 0;              // (wave-uniform) the waves that also sum dY's columns and take max |dY|
    if (r0 >= r1) { r0 = 0; r1 = 0; }                   // (chunk >= chunks: no trip; every load below is then out of range and reads 0)
    // one trip's products; `ok` masks what the pointer path (ConvTranspose1d) loaded from a clamped address
    auto products = [&](f32x4 (&a)[kU], float (&bv)[kU], unsigned ok) __attribute__((always_inline)) {
        if (TR) {
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                if (!((ok >> u) & 1u)) a[u] = zero4();
                if (!((ok >> (8 + u)) & 1u)) bv[u] = 0.0f;
            }
        }
        if (stats) {
#pragma unroll
            for (int u = 0; u < kU; ++u) {
#pragma unroll
                for (int e = 0; e < 4; ++e) amax_f = fmaxf(amax_f, fabsf(a[u][e]));    // (NaN-transparent enough: a NaN gradient shows up as NaN weights anyway)
                bsum = bsum + a[u];
            }
        }
#if ESMI_CHAIN_SPLIT
        const f32x4 b0 = {bv[0], bv[1], bv[2], bv[3]}, b1 = {bv[4], bv[5], bv[6], bv[7]};
        const bf16x3 b3 = split_bf16x3(b0, b1);
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const f32x4 a0 = {a[0][t4], a[1][t4], a[2][t4], a[3][t4]}, a1 = {a[4][t4], a[5][t4], a[6][t4], a[7][t4]};
            acc[t4] = mfma32_split(split_bf16x3(a0, a1), b3.hi, b3.mid, b3.lo, acc[t4]);
        }
#else
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) acc[t4] = mfma32(a[u][t4], bv[u], acc[t4]);
        }
#endif
    };
    // The loads.  Conv1d / Linear: bounds-checked buffer loads, so that nothing is clamped, compared or zeroed per value -- dY through
    // a resource that ends with the chunk's last row (rows past it read 0; a lane's byte offset is its first row's + a wave-uniform
    // multiple of the row size: one add per load), X through a resource over the whole tensor with the offset of (row, tap) kept
    // INCREMENTALLY from trip to trip (n_out >= 16: at most one utterance boundary per trip) and replaced by an out-of-range one where
    // the tap leaves the utterance or the row leaves the chunk.  (Round 5: the loop was VALU-bound on 64-bit address arithmetic, the
    // masks and the register rotation of the look-ahead: 640 VALU instructions, 66 of them quarter rate, per 24 MFMAs.)
    // ConvTranspose1d keeps the pointer path (wgrad_issue).
    constexpr unsigned kBig = 0x7FFFFFFFu;
    const BufRsrc r_dy = make_rsrc(dy + r0 * d.c_out, (r1 - r0) * (long)d.c_out * 4);
    const BufRsrc r_x = make_rsrc(x, (long)d.B * d.n_in * d.c_in * 4);
    const unsigned rowb = (unsigned)d.c_out * 4u, n = (unsigned)d.n_out;
    unsigned dy_off = vec_ok ? (unsigned)(lane_row * d.c_out + co4) * 4u : kBig;        // of this lane's first row of the next trip
    unsigned t_cur, x_off;                                                               // that row's position in its utterance, its X element
    {
        const unsigned rb = (unsigned)(r0 + lane_row), b0 = rb / n;
        t_cur = rb - b0 * n;
        x_off = (unsigned)(((long)b0 * d.n_in + (long)t_cur * d.stride + (j - d.pad)) * d.c_in + ci0 + i) * 4u;   // (mod 2^32 where the tap is outside)
    }
    const unsigned x_step = (unsigned)(d.stride * d.c_in) * 4u, x_wrap = (unsigned)((d.n_in - d.n_out * d.stride) * d.c_in) * 4u;
    long r_next = r0;                                                                    // first row of the next trip to be issued
    auto issue = [&](f32x4 (&a)[kU], float (&bv)[kU]) __attribute__((always_inline)) -> unsigned {
        if (TR) {
            const unsigned ok = wgrad_issue<kU, kS, TR>(d, x, dy, r_next + lane_row, r1 > r0 ? r1 : 1, j, co4, vec_ok, ci0 + i, ci_ok, a, bv);
            const bool any = r_next < r1;
            r_next += kTrip;
            return any ? ok : 0u;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) a[u] = buf_ld4(r_dy, dy_off + (unsigned)(kS * u) * rowb);
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const unsigned tu = t_cur + (unsigned)(kS * u);
            const bool wrap = tu >= n;
            const int ti = (int)mul24u(wrap ? tu - n : tu, (unsigned)d.stride) + (j - d.pad);
            const bool okx = ci_ok && (unsigned)ti < (unsigned)d.n_in && r_next + lane_row + kS * u < r1;
            const unsigned off = x_off + (unsigned)(kS * u) * x_step + (wrap ? x_wrap : 0u);
            bv[u] = buf_ld(r_x, okx ? off : kBig);
        }
        {   // this lane's first row of the trip after this one
            t_cur += (unsigned)kTrip;
            const bool wrap = t_cur >= n;
            t_cur = wrap ? t_cur - n : t_cur;
            x_off += (unsigned)kTrip * x_step + (wrap ? x_wrap : 0u);
            dy_off += (unsigned)kTrip * rowb;
            r_next += kTrip;
        }
        return 0x7FFFu;
    };
    // two register sets, each loaded one trip ahead of its products (no rotation: the loop body is two trips)
    f32x4 a0[kU], a1[kU];
    float v0[kU], v1[kU];
    unsigned ok0 = issue(a0, v0), ok1 = 0;
    for (long r = r0; r < r1; r += 2 * kTrip) {
        ok1 = issue(a1, v1);
        sched_fence();
        products(a0, v0, ok0);
        if (r + kTrip >= r1) break;
        ok0 = issue(a0, v0);
        sched_fence();
        products(a1, v1, ok1);
    }
    int amax_i = 0;
    if (dy_absmax && stats) {   // (wave-uniform) this wave's max |dY|; the workgroup's four are combined behind the barrier below
        amax_i = __builtin_bit_cast(int, amax_f) & 0x7FFFFFFF;
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) amax_i = max(amax_i, shfl_i(amax_i, lane ^ dd));
    }
    // ---- sum of the four waves' tiles: wave w0 writes [w0][reg][lane]; wave w then owns accumulator set t4 = w of the sum
    float* mine = red + (w * 64) * 64 + lane;
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[(16 * t4 + r) * 64] = acc[t4][r];
    }
    __syncthreads();
    f32x16 sum = zero16();
#pragma unroll
    for (int w0 = 0; w0 < kWgradWaves; ++w0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[r] += red[((w0 * 64) + 16 * w + r) * 64 + lane];
    }
    const long out_row = ygrp * pstride;                 // one partial row per workgroup
    const int ci = ci0 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = cb + 4 * tile_row(r, lane) + w;   // accumulator set t4 = w
        long wi;
        if (co < d.c_out && ci < d.c_in && conv_w_index(d, co, ci, j, &wi)) partial[out_row + wi] = sum[r];
    }
    if (dy_absmax && stats) {
        // ONE atomic per workgroup: 800 device-scope atomics on one address cost a decoder-size launch ~10 us (13 ns each, serialised)
        __syncthreads();
        int* ired = reinterpret_cast<int*>(red);
        if (lane == 0) ired[w] = amax_i;
        __syncthreads();
        if (w == 0 && lane == 0) {
            const int m = max(max(ired[0], ired[1]), max(ired[2], ired[3]));
            if (m > 0) atomicMax(dy_absmax, m);
        }
    }
    if (partial_bias && stats) {                         // bias: column sums of dY, the four waves' again added in wave order
        __syncthreads();
        float* bred = red;                                // [wave][128 channels]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = bsum[e] + shfl_xor_f(bsum[e], 32);      // even rows (kh = 0) + odd rows (kh = 1)
            if (kh == 0) bred[w * 128 + 4 * i + e] = v;
        }
        __syncthreads();
        if (w == 0 && lane < 64) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = lane + 64 * q;
                const float v = ((bred[c] + bred[128 + c]) + bred[256 + c]) + bred[384 + c];
                if (cb + c < d.c_out) partial_bias[out_row + cb + c] = v;
            }
        }
    }
}

// out[e] = sum over chunks of partial[chunk * stride + e], e < n.  64 elements x kReduceGroups interleaved chunk groups per
// workgroup; the group sums are added in group order (fixed order: reproducible).  (The CPU simulator build uses 4 groups and
// 8 loss workgroups: a fiber per thread makes 1024-thread workgroups the slowest thing in the test suite.)
// (kReduceGroups: wavesim_shim.h)
static __global__ __launch_bounds__(64 * kReduceGroups) void train_reduce_chunks_kernel(const float* __restrict__ partial, long n, long stride,
                                                                                long chunks, float* __restrict__ out,
                                                                                long n0 = -1, float* __restrict__ out1 = nullptr) {
    ESMI_DYN_LDS(red);   // 64 * kReduceGroups floats
    const int ex = (int)(threadIdx.x & 63), cy = (int)(threadIdx.x >> 6);
    const long q = (long)blockIdx.x * 64 + ex;
    float acc = 0.0f;
    if (q < n)
        for (long c = cy; c < chunks; c += kReduceGroups) acc += partial[c * stride + q];
    red[cy * 64 + ex] = acc;
    __syncthreads();
    if (cy == 0 && q < n) {
        float t = red[ex];
#pragma unroll
        for (int u = 1; u < kReduceGroups; ++u) t += red[u * 64 + ex];
        if (out1 && q >= n0) out1[q - n0] = t;      // second segment of a partial row (e.g. the bias sums after the weight sums)
        else out[q] = t;
    }
}
// tap-major GEMM copies of many weights in one launch (blockIdx.y = item): what pack_conv_kernel does per convolution
struct PackItem { const float* src; float* dst; int* zero_slot; int cout, cin, k, transposed, flip; };
constexpr int kPackBatch = 64;
struct PackBatch { int count; PackItem items[kPackBatch]; };
static __global__ void train_pack_batch_kernel(const PackBatch b) {
    const PackItem& it = b.items[blockIdx.y];
    if (it.zero_slot && blockIdx.x == 0 && threadIdx.x == 0) it.zero_slot[0] = 0;
    const long n = (long)it.cout * it.cin * it.k;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int i = (int)(e % it.cin);
        const int o = (int)((e / it.cin) % it.cout);
        const int j = (int)(e / ((long)it.cin * it.cout));
        const int js = it.flip ? it.k - 1 - j : j;
        it.dst[e] = it.transposed ? it.src[((long)i * it.cout + o) * it.k + js] : it.src[((long)o * it.cin + i) * it.k + js];
    }
}
// the same for a batch of independent reductions in one launch (blockIdx.y = item): the queued second stages of a training step
struct ReduceItem { const float* partial; long n, stride, chunks; float* out; long n0; float* out1; };
constexpr int kReduceBatch = 48;
// (first[i]: the first workgroup of item i in the launch's 1-D grid, first[count]: the grid size.  A 2-D grid sized for the largest
// item launched ~14,700 workgroups of 16 waves for the step's 48 reductions, most of them leaving at once: 57 us, bound by the dispatcher.)
struct ReduceBatch { int count; int first[kReduceBatch + 1]; ReduceItem items[kReduceBatch]; };
static __global__ __launch_bounds__(64 * kReduceGroups) void train_reduce_batch_kernel(const ReduceBatch b) {
    ESMI_DYN_LDS(red);   // 64 * kReduceGroups floats
    int ii = 0;
    while (ii + 1 < b.count && (int)blockIdx.x >= b.first[ii + 1]) ++ii;     // (workgroup-uniform)
    const ReduceItem& it = b.items[ii];
    const int ex = (int)(threadIdx.x & 63), cy = (int)(threadIdx.x >> 6);
    const long q = (long)((int)blockIdx.x - b.first[ii]) * 64 + ex;
    float acc = 0.0f;
    if (q < it.n)
        for (long c = cy; c < it.chunks; c += 4 * kReduceGroups) {   // four loads in flight, added in chunk order (the loop was one exposed
            float v[4];                                               // round trip per partial row: 57 us for the step's 55 MB)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long cc = c + (long)u * kReduceGroups;
                v[u] = it.partial[(cc < it.chunks ? cc : c) * it.stride + q];
                if (cc >= it.chunks) v[u] = 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u];
        }
    red[cy * 64 + ex] = acc;
    __syncthreads();
    if (cy == 0 && q < it.n) {
        float t = red[ex];
#pragma unroll
        for (int u = 1; u < kReduceGroups; ++u) t += red[u * 64 + ex];
        if (it.out1 && q >= it.n0) it.out1[q - it.n0] = t;
        else it.out[q] = t;
    }
}
// partial[chunk][c] = sum over the chunk's rows of v[row, c]   (bias gradients)
static __global__ void train_colsum_kernel(const float* __restrict__ v, long rows, int C, float* __restrict__ partial, long pstride, int chunk) {
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (c >= C) return;
    const long r0 = (long)blockIdx.y * chunk, r1 = r0 + chunk < rows ? r0 + chunk : rows;
    float acc = 0.0f;
    for (long r = r0; r < r1; ++r) acc += v[r * C + c];
    partial[(long)blockIdx.y * pstride + c] = acc;
}

// ---- LayerNorm over the last dim (biased variance, eps inside the sqrt): one wave per row, lanes across the channels
// (coalesced), sums by a fixed butterfly; mean / rstd kept for backward
__device__ __forceinline__ float ln_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += shfl_xor_f(v, m);
    return v;
}
// res != NULL: the normalised tensor is x + res, written to `xsum` for the backward (LN(f(x) + x), networks.py:75,83,301);
// rowmask[r] != 0: the output row is zero (masked_fill after the norm, networks.py:76,84) -- the backward kernels take the same mask
// (x and y are NOT __restrict__: the two-launch conv + LayerNorm fallback of train.py runs the norm in place, x == y; every lane reads
// its elements of a row before it writes them)
static __global__ __launch_bounds__(256) void train_ln_fwd_kernel(const float* x, const float* __restrict__ g,
                                                           const float* __restrict__ b, long rows, int C, float eps,
                                                           float* y, float* __restrict__ mean, float* __restrict__ rstd,
                                                           const float* __restrict__ res, float* __restrict__ xsum,
                                                           const unsigned char* __restrict__ rowmask, int relu_out) {
    const long r = (long)blockIdx.x * 4 + wave_id();
    if (r >= rows) return;
    const int lane = lane_id();
    const float* xr = x + r * C;
    float m = 0.0f;
    if (res) {   // this lane re-reads only what it wrote
        for (int c = lane; c < C; c += 64) { const float v = xr[c] + res[r * C + c]; xsum[r * C + c] = v; m += v; }
        xr = xsum + r * C;
    } else {
        for (int c = lane; c < C; c += 64) m += xr[c];
    }
    m = ln_wave_sum(m) / (float)C;
    float v = 0.0f;
    for (int c = lane; c < C; c += 64) { const float dlt = xr[c] - m; v = fmaf(dlt, dlt, v); }
    const float rs = 1.0f / sqrtf(ln_wave_sum(v) / (float)C + eps);
    if (lane == 0) { mean[r] = m; rstd[r] = rs; }
    const bool masked = rowmask && rowmask[r];
    for (int c = lane; c < C; c += 64) {
        const float v = fmaf((xr[c] - m) * rs, g[c], b[c]);
        y[r * C + c] = masked ? 0.0f : (relu_out ? fmaxf(v, 0.0f) : v);
    }
}
// The same for C = 4 LPR in {32, 64, 128, 256} on 16-byte aligned tensors: LPR lanes per row hold the row in registers (one read of
// x instead of three, 16-byte accesses), 64 / LPR rows per wave.  (The scalar form moves 78 MB in 30 us at B = 128.)
template <int LPR>
static __global__ __launch_bounds__(256) void train_ln_fwd4_kernel(const float* x, const float* __restrict__ g,
                                                            const float* __restrict__ b, long rows, float eps,
                                                            float* y, float* __restrict__ mean, float* __restrict__ rstd,
                                                            const float* __restrict__ res, float* __restrict__ xsum,
                                                            const unsigned char* __restrict__ rowmask, int relu_out) {
    constexpr int RPW = 64 / LPR, C = 4 * LPR;
    const int lane = lane_id(), sub = lane / LPR, l = lane % LPR;
    const long r = ((long)blockIdx.x * 4 + wave_id()) * RPW + sub;
    const bool live = r < rows;                        // (no early exit: the row sums are wave collectives)
    const long rc = live ? r : rows - 1;
    f32x4 v = ld4(x + rc * C + 4 * l);
    if (res) {
        v = v + ld4(res + rc * C + 4 * l);
        if (live) *reinterpret_cast<f32x4*>(xsum + rc * C + 4 * l) = v;
    }
    float m = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
    for (int k = LPR / 2; k > 0; k >>= 1) m += shfl_xor_f(m, k);
    m = m / (float)C;
    f32x4 dl;
#pragma unroll
    for (int e = 0; e < 4; ++e) dl[e] = v[e] - m;
    float q = fmaf(dl[3], dl[3], fmaf(dl[2], dl[2], fmaf(dl[1], dl[1], dl[0] * dl[0])));
#pragma unroll
    for (int k = LPR / 2; k > 0; k >>= 1) q += shfl_xor_f(q, k);
    const float rs = 1.0f / sqrtf(q / (float)C + eps);
    if (l == 0 && live) { mean[r] = m; rstd[r] = rs; }
    const bool masked = rowmask && rowmask[rc];
    const f32x4 gv = ld4(g + 4 * l), bv = ld4(b + 4 * l);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t = fmaf(dl[e] * rs, gv[e], bv[e]);
        o[e] = masked ? 0.0f : (relu_out ? fmaxf(t, 0.0f) : t);
    }
    if (live) *reinterpret_cast<f32x4*>(y + rc * C + 4 * l) = o;
}
// derivative of ReLU / tanh from the activation's OUTPUT y (kind as esmi_dev.h Act; 0: 1)
__device__ __forceinline__ float act_grad_from_output(int kind, float y) {
    return kind == ACT_RELU ? (y > 0.0f ? 1.0f : 0.0f) : (kind == ACT_TANH ? 1.0f - y * y : 1.0f);
}
// in_act: x is the output of that activation (LN(tanh(conv)), networks.py:298; LN(relu(conv)), :152-153) and dx is returned for the
// PRE-activation tensor -- the activation's backward rides in this launch
static __global__ __launch_bounds__(256) void train_ln_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ dy, long rows, int C, float* __restrict__ dx,
                                                              const unsigned char* __restrict__ rowmask, int in_act,
                                                              const float* __restrict__ y_relu) {
    const long r = (long)blockIdx.x * 4 + wave_id();
    if (r >= rows) return;
    const int lane = lane_id();
    if (rowmask && rowmask[r]) {   // the forward zeroed this row: no gradient passes
        for (int c = lane; c < C; c += 64) dx[r * C + c] = 0.0f;
        return;
    }
    const float m = mean[r], rs = rstd[r];
    float s1 = 0.0f, s2 = 0.0f;
    for (int c = lane; c < C; c += 64) {
        const float dyv = (y_relu && !(y_relu[r * C + c] > 0.0f)) ? 0.0f : dy[r * C + c];
        const float xh = (x[r * C + c] - m) * rs, dh = dyv * g[c];
        s1 += dh;
        s2 = fmaf(dh, xh, s2);
    }
    s1 = ln_wave_sum(s1) / (float)C;
    s2 = ln_wave_sum(s2) / (float)C;
    for (int c = lane; c < C; c += 64) {
        const float dyv = (y_relu && !(y_relu[r * C + c] > 0.0f)) ? 0.0f : dy[r * C + c];
        const float xv = x[r * C + c], xh = (xv - m) * rs, dh = dyv * g[c];
        dx[r * C + c] = rs * (dh - s1 - xh * s2) * act_grad_from_output(in_act, xv);
    }
}
// dx and the parameter-gradient partials in one pass (C <= 256): a 4-wave workgroup per kLnRows rows, one wave per quarter, lanes
// across the channels; every lane keeps dgamma / dbeta sums of its (up to four) channels over its wave's rows, the four waves'
// sums are added in wave order through LDS -> partial[workgroup][dg (C) | db (C)]
constexpr int kLnRows = 64;
static __global__ __launch_bounds__(256) void train_ln_bwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ dy, long rows, int C,
                                                                 float* __restrict__ dx, float* __restrict__ partial,
                                                                 const unsigned char* __restrict__ rowmask, int in_act,
                                                                 const float* __restrict__ y_relu) {
    ESMI_DYN_LDS(red);   // [4 waves][2][256] floats
    const long chunk = blockIdx.x;
    const int lane = lane_id(), w = wave_id();
    const long r0 = chunk * kLnRows + w * (kLnRows / 4);
    const long rend = chunk * kLnRows + kLnRows < rows ? chunk * kLnRows + kLnRows : rows;
    const long r1 = r0 + kLnRows / 4 < rend ? r0 + kLnRows / 4 : rend;
    float gg[4], dga[4] = {0.f, 0.f, 0.f, 0.f}, dba[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) gg[u] = lane + 64 * u < C ? g[lane + 64 * u] : 0.0f;
    for (long r = r0; r < r1; ++r) {
        const float m = mean[r], rs = rstd[r];
        const bool masked = rowmask && rowmask[r];   // (a zero dy row: dx = 0, no contribution to dgamma / dbeta)
        float xv[4], xh[4], dh[4], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = lane + 64 * u;
            const bool ok = c < C;
            const float d = (ok && !masked && !(y_relu && !(y_relu[r * C + c] > 0.0f))) ? dy[r * C + c] : 0.0f;
            xv[u] = ok ? x[r * C + c] : 0.0f;
            xh[u] = ok ? (xv[u] - m) * rs : 0.0f;
            dh[u] = d * gg[u];
            s1 += dh[u];
            s2 = fmaf(dh[u], xh[u], s2);
            dga[u] = fmaf(d, xh[u], dga[u]);
            dba[u] += d;
        }
        s1 = ln_wave_sum(s1) / (float)C;
        s2 = ln_wave_sum(s2) / (float)C;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = lane + 64 * u;
            if (c < C) dx[r * C + c] = rs * (dh[u] - s1 - xh[u] * s2) * act_grad_from_output(in_act, xv[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        red[(w * 2 + 0) * 256 + lane + 64 * u] = dga[u];
        red[(w * 2 + 1) * 256 + lane + 64 * u] = dba[u];
    }
    __syncthreads();
    for (int e = (int)threadIdx.x; e < 2 * C; e += 256) {      // e < C: dgamma[e], else dbeta[e - C]
        const int which = e >= C ? 1 : 0, c = e - which * C;
        partial[chunk * 2 * C + e] = ((red[(0 * 2 + which) * 256 + c] + red[(1 * 2 + which) * 256 + c]) + red[(2 * 2 + which) * 256 + c]) +
                                     red[(3 * 2 + which) * 256 + c];
    }
}

// The same for C = 4 LPR in {32, 64, 128} on 16-byte aligned tensors (round 5): LPR lanes per row, 64 / LPR rows per pass, 16-byte
// accesses, and ALL of a wave's 16 rows loaded before the first is used -- the kernel above walks its rows one by one, every row a
// dependent round trip (mean, rstd, mask, then x and dy): 20 us for an encoder-size launch of 1.6 MB, 29 us at decoder size.  A lane's
// dgamma / dbeta sums run over the rows of its lane group; the groups are then added by a fixed butterfly, the waves in wave order.
template <int LPR>
static __global__ __launch_bounds__(256) void train_ln_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ dy, long rows, float* __restrict__ dx,
                                                            float* __restrict__ partial, const unsigned char* __restrict__ rowmask,
                                                            int in_act, const float* __restrict__ y_relu) {
    constexpr int RPW = 64 / LPR, C = 4 * LPR, WROWS = kLnRows / 4, NP = WROWS / RPW;
    ESMI_DYN_LDS(red);   // [4 waves][2][256] floats
    const long chunk = blockIdx.x;
    const int lane = lane_id(), w = wave_id(), sub = lane / LPR, l = lane % LPR;
    const long rw0 = chunk * kLnRows + (long)w * WROWS + sub;          // this lane's row of pass p: rw0 + RPW p
    const f32x4 gv = ld4(g + 4 * l);
    f32x4 xv[NP], dv[NP];
    float mm[NP], rr[NP];
    unsigned livem = 0u, maskm = 0u;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const long r = rw0 + RPW * p, rc = r < rows ? r : rows - 1;
        xv[p] = ld4(x + rc * C + 4 * l);
        dv[p] = ld4(dy + rc * C + 4 * l);
        mm[p] = mean[rc];
        rr[p] = rstd[rc];
        if (r < rows) livem |= 1u << p;
    }
    if (rowmask) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const long r = rw0 + RPW * p;
            if (rowmask[r < rows ? r : rows - 1]) maskm |= 1u << p;
        }
    }
    if (y_relu) {   // dy counts only where the forward's relu(LN(.)) output is positive
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const long r = rw0 + RPW * p;
            const f32x4 yv = ld4(y_relu + (r < rows ? r : rows - 1) * C + 4 * l);
#pragma unroll
            for (int e = 0; e < 4; ++e) if (!(yv[e] > 0.0f)) dv[p][e] = 0.0f;
        }
    }
    f32x4 dga = zero4(), dba = zero4();
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const bool live = (livem >> p) & 1u;
        if (!live || ((maskm >> p) & 1u)) dv[p] = zero4();            // (a zero dy row: dx = 0, no contribution to dgamma / dbeta)
        f32x4 xh, dh;
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xh[e] = (xv[p][e] - mm[p]) * rr[p];
            dh[e] = dv[p][e] * gv[e];
            s1 += dh[e];
            s2 = fmaf(dh[e], xh[e], s2);
            dga[e] = fmaf(dv[p][e], xh[e], dga[e]);
            dba[e] += dv[p][e];
        }
#pragma unroll
        for (int k = LPR / 2; k > 0; k >>= 1) { s1 += shfl_xor_f(s1, k); s2 += shfl_xor_f(s2, k); }
        s1 = s1 / (float)C;
        s2 = s2 / (float)C;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rr[p] * (dh[e] - s1 - xh[e] * s2) * act_grad_from_output(in_act, xv[p][e]);
        if (live) *reinterpret_cast<f32x4*>(dx + (rw0 + RPW * p) * C + 4 * l) = o;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int k = LPR; k < 64; k <<= 1) { dga[e] += shfl_xor_f(dga[e], k); dba[e] += shfl_xor_f(dba[e], k); }
    }
    if (sub == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[(w * 2 + 0) * 256 + 4 * l + e] = dga[e];
            red[(w * 2 + 1) * 256 + 4 * l + e] = dba[e];
        }
    }
    __syncthreads();
    for (int e = (int)threadIdx.x; e < 2 * C; e += 256) {      // e < C: dgamma[e], else dbeta[e - C]
        const int which = e >= C ? 1 : 0, c = e - which * C;
        partial[chunk * 2 * C + e] = ((red[(0 * 2 + which) * 256 + c] + red[(1 * 2 + which) * 256 + c]) + red[(2 * 2 + which) * 256 + c]) +
                                     red[(3 * 2 + which) * 256 + c];
    }
}

// partial[chunk][0][c] = sum dy * xhat, partial[chunk][1][c] = sum dy over the chunk's rows
static __global__ void train_ln_bwd_params_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                           const float* __restrict__ dy, long rows, int C, float* __restrict__ partial,
                                           const unsigned char* __restrict__ rowmask, const float* __restrict__ y_relu) {
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (c >= C) return;
    const long r0 = (long)blockIdx.y * kTrainChunk, r1 = r0 + kTrainChunk < rows ? r0 + kTrainChunk : rows;
    float a = 0.0f, s = 0.0f;
    for (long r = r0; r < r1; ++r) {
        const float d = ((rowmask && rowmask[r]) || (y_relu && !(y_relu[r * C + c] > 0.0f))) ? 0.0f : dy[r * C + c];
        a = fmaf(d, (x[r * C + c] - mean[r]) * rstd[r], a);
        s += d;
    }
    partial[(long)blockIdx.y * 2 * C + c] = a;
    partial[(long)blockIdx.y * 2 * C + C + c] = s;
}

// ---- activations: kind as esmi_dev.h Act (1 ReLU, 2 GELU erf, 3 tanh); backward reads y for ReLU / tanh and x for GELU
static __global__ void train_act_fwd_kernel(const float* __restrict__ x, long n, int kind, float* __restrict__ y) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const float v = x[q];
    y[q] = kind == ACT_RELU ? fmaxf(v, 0.0f) : (kind == ACT_GELU ? 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)) : tanhf(v));
}
static __global__ void train_act_bwd_kernel(const float* __restrict__ saved, const float* __restrict__ dy, long n, int kind,
                                     float* __restrict__ dx) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const float s = saved[q];
    float d;
    if (kind == ACT_RELU) d = s > 0.0f ? 1.0f : 0.0f;
    else if (kind == ACT_TANH) d = 1.0f - s * s;
    else d = 0.5f * (1.0f + erff(s * 0.70710678118654752f)) + s * 0.3989422804014327f * expf(-0.5f * s * s);
    dx[q] = dy[q] * d;
}

// ---- attention core, blocks.py:43-64 (scores are NOT masked there): qkv (B, N, 3, h, C) -> P (B, h, N, N), ctx (B, N, h*C).
// One 64-thread workgroup per (b, head, query row): lanes share the keys for the scores (wave reductions for max / sum in a
// fixed butterfly order), then the channels for the context row.
__device__ __forceinline__ float wave_allreduce_max(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = fmaxf(v, shfl_xor_f(v, m));
    return v;
}
__device__ __forceinline__ float wave_allreduce_sum(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += shfl_xor_f(v, m);
    return v;
}
static __global__ __launch_bounds__(64) void train_attn_fwd_kernel(const float* __restrict__ qkv, int B, int N, int C, int h, float scale,
                                                            float* __restrict__ P, float* __restrict__ ctx) {
    const long q = blockIdx.x;
    const int lane = (int)threadIdx.x;
    const int i = (int)(q % N), hd = (int)((q / N) % h), b = (int)(q / ((long)N * h));
    const long ld = 3L * h * C;
    const float* qi = qkv + ((long)b * N + i) * ld + (long)hd * C;
    float* p = P + (((long)b * h + hd) * N + i) * N;
    float mx = -3.0e38f;
    for (int j = lane; j < N; j += 64) {
        const float* kj = qkv + ((long)b * N + j) * ld + (long)(h + hd) * C;
        float s = 0.0f;
        for (int c = 0; c < C; ++c) s = fmaf(qi[c], kj[c], s);
        s *= scale;
        p[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_allreduce_max(mx);
    float sum = 0.0f;
    for (int j = lane; j < N; j += 64) { const float e = expf(p[j] - mx); p[j] = e; sum += e; }
    sum = wave_allreduce_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < N; j += 64) p[j] *= inv;
    __syncthreads();                                   // the row of P is complete for every lane
    float* o = ctx + ((long)b * N + i) * h * C + (long)hd * C;
    for (int c = lane; c < C; c += 64) {
        float a = 0.0f;
        for (int j = 0; j < N; ++j) a = fmaf(p[j], qkv[((long)b * N + j) * ld + (long)(2 * h + hd) * C + c], a);
        o[c] = a;
    }
}
// row i: dP = dctx_i . V^T, dS = P o (dP - <P, dP>), dq_i = scale dS K;  dS (B, h, N, N) kept for the column pass
static __global__ __launch_bounds__(64) void train_attn_bwd_rows_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                                 const float* __restrict__ dctx, int B, int N, int C, int h,
                                                                 float scale, float* __restrict__ dS, float* __restrict__ dqkv) {
    const long q = blockIdx.x;
    const int lane = (int)threadIdx.x;
    const int i = (int)(q % N), hd = (int)((q / N) % h), b = (int)(q / ((long)N * h));
    const long ld = 3L * h * C;
    const float* p = P + (((long)b * h + hd) * N + i) * N;
    float* ds = dS + (((long)b * h + hd) * N + i) * N;
    const float* go = dctx + ((long)b * N + i) * h * C + (long)hd * C;
    float dot = 0.0f;
    for (int j = lane; j < N; j += 64) {
        const float* vj = qkv + ((long)b * N + j) * ld + (long)(2 * h + hd) * C;
        float a = 0.0f;
        for (int c = 0; c < C; ++c) a = fmaf(go[c], vj[c], a);
        ds[j] = a;
        dot = fmaf(p[j], a, dot);
    }
    dot = wave_allreduce_sum(dot);
    for (int j = lane; j < N; j += 64) ds[j] = p[j] * (ds[j] - dot);
    __syncthreads();
    float* dq = dqkv + ((long)b * N + i) * ld + (long)hd * C;
    for (int c = lane; c < C; c += 64) {
        float a = 0.0f;
        for (int j = 0; j < N; ++j) a = fmaf(ds[j], qkv[((long)b * N + j) * ld + (long)(h + hd) * C + c], a);
        dq[c] = a * scale;
    }
}
// The same two kernels with the head's K and V staged once in LDS ([N][C + 1] each: a lane per key reads its row at an odd
// stride, a lane per channel reads consecutive words -- both conflict-free), one 256-thread workgroup per (b, head), every wave
// taking query rows w, w + 4, ...  Used whenever 2 N (C + 1) + 4 N floats fit (tiny / small ES at any realistic length); the
// global-memory versions above remain for the rest.
__host__ __device__ inline size_t train_attn_lds_bytes(int N, int C) { return ((size_t)2 * N * (C + 1) + 4 * (size_t)N) * sizeof(float); }
static __global__ __launch_bounds__(256) void train_attn_fwd_lds_kernel(const float* __restrict__ qkv, int B, int N, int C, int h, float scale,
                                                                 float* __restrict__ P, float* __restrict__ ctx) {
    ESMI_DYN_LDS(lds);
    float* Ks = lds;
    float* Vs = lds + (long)N * (C + 1);
    float* prow = Vs + (long)N * (C + 1) + (long)wave_id() * N;      // this wave's softmax row
    const int hd = (int)(blockIdx.x % h), b = (int)(blockIdx.x / h), lane = lane_id(), w = wave_id();
    const long ld = 3L * h * C;
    for (int e = (int)threadIdx.x; e < N * C; e += 256) {
        const int j = e / C, c = e - j * C;
        Ks[j * (C + 1) + c] = qkv[((long)b * N + j) * ld + (long)(h + hd) * C + c];
        Vs[j * (C + 1) + c] = qkv[((long)b * N + j) * ld + (long)(2 * h + hd) * C + c];
    }
    __syncthreads();
    // gridDim.y row segments per (utterance, head): B * h workgroups alone leave most of the chip idle at the reference's batch size
    const int seg = (N + (int)gridDim.y - 1) / (int)gridDim.y, i_lo = (int)blockIdx.y * seg, i_hi = i_lo + seg < N ? i_lo + seg : N;
    for (int i = i_lo + w; i < i_hi; i += 4) {
        const float* qi = qkv + ((long)b * N + i) * ld + (long)hd * C;
        float* p = P + (((long)b * h + hd) * N + i) * N;
        float mx = -3.0e38f;
        for (int j = lane; j < N; j += 64) {
            float s = 0.0f;
            for (int c = 0; c < C; ++c) s = fmaf(qi[c], Ks[j * (C + 1) + c], s);
            s *= scale;
            prow[j] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_allreduce_max(mx);
        float sum = 0.0f;
        for (int j = lane; j < N; j += 64) { const float e = expf(prow[j] - mx); prow[j] = e; sum += e; }
        sum = wave_allreduce_sum(sum);
        const float inv = 1.0f / sum;
        for (int j = lane; j < N; j += 64) { const float v = prow[j] * inv; prow[j] = v; p[j] = v; }
        lds_wave_sync();                               // the row is read back by other lanes of this wave
        float* o = ctx + ((long)b * N + i) * h * C + (long)hd * C;
        for (int c = lane; c < C; c += 64) {
            float a = 0.0f;
            for (int j = 0; j < N; ++j) a = fmaf(prow[j], Vs[j * (C + 1) + c], a);
            o[c] = a;
        }
        lds_wave_sync();                               // before the next row overwrites prow
    }
}
static __global__ __launch_bounds__(256) void train_attn_bwd_rows_lds_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                                      const float* __restrict__ dctx, int B, int N, int C, int h,
                                                                      float scale, float* __restrict__ dS, float* __restrict__ dqkv) {
    ESMI_DYN_LDS(lds);
    float* Ks = lds;
    float* Vs = lds + (long)N * (C + 1);
    float* drow = Vs + (long)N * (C + 1) + (long)wave_id() * N;
    const int hd = (int)(blockIdx.x % h), b = (int)(blockIdx.x / h), lane = lane_id(), w = wave_id();
    const long ld = 3L * h * C;
    for (int e = (int)threadIdx.x; e < N * C; e += 256) {
        const int j = e / C, c = e - j * C;
        Ks[j * (C + 1) + c] = qkv[((long)b * N + j) * ld + (long)(h + hd) * C + c];
        Vs[j * (C + 1) + c] = qkv[((long)b * N + j) * ld + (long)(2 * h + hd) * C + c];
    }
    __syncthreads();
    // gridDim.y row segments per (utterance, head): B * h workgroups alone leave most of the chip idle at the reference's batch size
    const int seg = (N + (int)gridDim.y - 1) / (int)gridDim.y, i_lo = (int)blockIdx.y * seg, i_hi = i_lo + seg < N ? i_lo + seg : N;
    for (int i = i_lo + w; i < i_hi; i += 4) {
        const float* p = P + (((long)b * h + hd) * N + i) * N;
        float* ds = dS + (((long)b * h + hd) * N + i) * N;
        const float* go = dctx + ((long)b * N + i) * h * C + (long)hd * C;
        float dot = 0.0f;
        for (int j = lane; j < N; j += 64) {
            float a = 0.0f;
            for (int c = 0; c < C; ++c) a = fmaf(go[c], Vs[j * (C + 1) + c], a);
            drow[j] = a;
            dot = fmaf(p[j], a, dot);
        }
        dot = wave_allreduce_sum(dot);
        for (int j = lane; j < N; j += 64) { const float v = p[j] * (drow[j] - dot); drow[j] = v; ds[j] = v; }
        lds_wave_sync();
        float* dq = dqkv + ((long)b * N + i) * ld + (long)hd * C;
        for (int c = lane; c < C; c += 64) {
            float a = 0.0f;
            for (int j = 0; j < N; ++j) a = fmaf(drow[j], Ks[j * (C + 1) + c], a);
            dq[c] = a * scale;
        }
        lds_wave_sync();
    }
}

// column j, channel c: dk_j = scale dS^T Q, dv_j = P^T dctx   (one thread per (b, head, j, c))
static __global__ void train_attn_bwd_cols_kernel(const float* __restrict__ qkv, const float* __restrict__ P, const float* __restrict__ dS,
                                           const float* __restrict__ dctx, int B, int N, int C, int h, float scale,
                                           float* __restrict__ dqkv) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long)B * h * N * C) return;
    const int c = (int)(q % C), j = (int)((q / C) % N), hd = (int)((q / ((long)C * N)) % h), b = (int)(q / ((long)C * N * h));
    const long ld = 3L * h * C;
    float a = 0.0f, v = 0.0f;
    for (int i = 0; i < N; ++i) {
        const long pi = (((long)b * h + hd) * N + i) * N + j;
        a = fmaf(dS[pi], qkv[((long)b * N + i) * ld + (long)hd * C + c], a);
        v = fmaf(P[pi], dctx[((long)b * N + i) * h * C + (long)hd * C + c], v);
    }
    dqkv[((long)b * N + j) * ld + (long)(h + hd) * C + c] = a * scale;
    dqkv[((long)b * N + j) * ld + (long)(2 * h + hd) * C + c] = v;
}

// ---- embedding: forward gather (out-of-range ids read row 0); backward one thread per table element, rows with id ==
// padding_idx get no gradient (nn.Embedding(padding_idx), networks.py:32)
static __global__ void train_embed_fwd_kernel(const int* __restrict__ ids, const float* __restrict__ table, long rows, int V, int C,
                                       float* __restrict__ out) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= rows * C) return;
    int id = ids[q / C];
    if (id < 0 || id >= V) id = 0;
    out[q] = table[(long)id * C + q % C];
}
// (blockDim = 64.)  The chunk's ids come in by coalesced loads, 64 at a time; one ballot per vocabulary entry of the wave (its 64
// consecutive (v, c) elements span 64 / C + 1 entries at most) gives every lane the bit mask of ITS entry's rows, and the lane then
// visits only those, in row order: the same sums in the same order as a compare per (element, row) -- which read ids[r] through the
// scalar cache once per row (256 dependent ~200 ns loads per thread: 59 us for the phoneme table at B = 128).
static __global__ __launch_bounds__(64) void train_embed_bwd_kernel(const int* __restrict__ ids, const float* __restrict__ dy, long rows, int V, int C,
                                       int padding_idx, float* __restrict__ partial) {
    const long q0 = (long)blockIdx.x * 64, q = q0 + threadIdx.x;   // partial[chunk][v][c] over the chunk's rows
    const long nq = (long)V * C;
    const bool live = q < nq;
    const int v = live ? (int)(q / C) : -1, c = live ? (int)(q % C) : 0, lane = lane_id();
    const bool take = live && v != padding_idx;
    const long qe = q0 + 63 < nq ? q0 + 63 : nq - 1;
    const int v_lo = (int)(q0 / C), v_hi = (int)(qe / C);           // wave-uniform
    const long r0 = (long)blockIdx.y * kTrainChunk, r1 = r0 + kTrainChunk < rows ? r0 + kTrainChunk : rows;
    float acc = 0.0f;
    for (long base = r0; base < r1; base += 64) {
        const int idv = base + lane < r1 ? ids[base + lane] : -1;
        unsigned long long mine = 0ull;
        for (int vv = v_lo; vv <= v_hi; ++vv) {
            const unsigned long long m = ballot64(idv == vv);
            if (vv == v) mine = m;
        }
        if (!take) mine = 0ull;
        while (mine) {                                              // ascending rows
            const int rr = __builtin_ctzll(mine);
            mine &= mine - 1;
            acc += dy[(base + rr) * C + c];
        }
    }
    if (live) partial[(long)blockIdx.y * nq + q] = acc;
}

// ---- row masking (masked_fill(mask, 0) with a per-row mask), residual add, column-block copy (torch.cat / its gradient)
static __global__ void train_mask_rows_kernel(const float* __restrict__ x, const unsigned char* __restrict__ mask, long rows, int C,
                                       float* __restrict__ y) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= rows * C) return;
    y[q] = mask[q / C] ? 0.0f : x[q];
}
static __global__ void train_add_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, float* __restrict__ y) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) y[q] = a[q] + b[q];
}
static __global__ void train_copy_cols_kernel(const float* __restrict__ src, int ld_src, int col_src, float* __restrict__ dst, int ld_dst,
                                       int col_dst, long rows, int C) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= rows * C) return;
    const long r = q / C;
    const int c = (int)(q % C);
    dst[r * ld_dst + col_dst + c] = src[r * ld_src + col_src + c];
}

// torch.cat(parts, dim=-1) of up to 8 row-major parts in ONE launch, or its backward (the slices of dcat back into the parts);
// parts whose bit is set in `masked` are taken as zero on rows with rowmask[r] != 0 (x.masked_fill(mask, 0) before the cat,
// networks.py:366-368) -- the same mask zeroes those rows of their gradient
struct CatArgs { float* part[8]; int width[8]; int col[8]; int n, tot; long rows; float* cat; const unsigned char* rowmask; unsigned masked; int backward; };
static __global__ void train_cat_kernel(const CatArgs a) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.rows * a.tot) return;
    const long r = q / a.tot;
    const int c = (int)(q - r * a.tot);
    int pi = 0;
#pragma unroll
    for (int j = 1; j < 8; ++j) pi += (j < a.n && c >= a.col[j]) ? 1 : 0;
    float* part = a.part[0];
    int w = a.width[0], c0 = a.col[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) if (pi == j) { part = a.part[j]; w = a.width[j]; c0 = a.col[j]; }
    const bool zero = ((a.masked >> pi) & 1u) && a.rowmask && a.rowmask[r];
    if (a.backward) part[r * w + (c - c0)] = zero ? 0.0f : a.cat[q];
    else a.cat[q] = zero ? 0.0f : part[r * w + (c - c0)];
}

// ---- length regulator (networks.py:233-244) forward / backward on the inclusive duration cumsum `cum` (B, T):
// frame f of utterance b copies phoneme t with cum[t-1] <= f < cum[t]; frames >= cum[T-1] are zero padding
static __global__ void train_repeat_fwd_kernel(const float* __restrict__ feat, const int* __restrict__ cum, int B, int T, int C, int L,
                                        float* __restrict__ out) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long)B * L * C) return;
    const int c = (int)(q % C), f = (int)((q / C) % L), b = (int)(q / ((long)C * L));
    const int* cb = cum + (long)b * T;
    int lo = 0, hi = T;                       // first t with cum[t] > f
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cb[mid] > f) hi = mid; else lo = mid + 1; }
    out[q] = lo < T ? feat[((long)b * T + lo) * C + c] : 0.0f;
}
// the same, four channels per thread (C % 4 == 0, 16-byte aligned tensors, B L C / 4 < 2^31): one search per 16 bytes instead of one per float
static __global__ void train_repeat_fwd4_kernel(const float* __restrict__ feat, const int* __restrict__ cum, int B, int T, int C4, int L,
                                         float* __restrict__ out) {
    const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (unsigned)B * (unsigned)L * (unsigned)C4) return;
    const unsigned row = q / (unsigned)C4, c4 = q - row * (unsigned)C4, b = row / (unsigned)L, f = row - b * (unsigned)L;
    const int* cb = cum + (long)b * T;
    int lo = 0, hi = T;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cb[mid] > (int)f) hi = mid; else lo = mid + 1; }
    f32x4 v = zero4();
    if (lo < T) v = ld4(feat + (((long)b * T + lo) * C4 + c4) * 4);
    *reinterpret_cast<f32x4*>(out + (long)q * 4) = v;
}
static __global__ void train_repeat_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ cum, int B, int T, int C, int L,
                                        float* __restrict__ dfeat) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long)B * T * C) return;
    const int c = (int)(q % C), t = (int)((q / C) % T), b = (int)(q / ((long)C * T));
    const int f0 = t ? cum[(long)b * T + t - 1] : 0;
    int f1 = cum[(long)b * T + t];
    f1 = f1 < L ? f1 : L;
    float acc = 0.0f;
    for (int f = f0; f < f1; ++f) acc += dout[((long)b * L + f) * C + c];
    dfeat[q] = acc;
}

// ---- the loss of model.py:167-216 and its gradient seeds, in two deterministic stages: kLossBlocks workgroups write partial
// sums (counts, mel |d|, three squared errors) in a fixed order, one workgroup adds them up; the gradient kernel then scales.
//   out[0..3] = mel L1, pitch MSE, energy MSE, log-duration MSE (means over the unmasked elements); out[4] = 10 a + 2 b + 2 c + d
struct LossP {
    const float *mel_pred, *mel, *pitch_pred, *pitch, *energy_pred, *energy, *dur_pred;
    const int* dur;
    const unsigned char *mel_mask, *ph_mask;   // 1 = padding; NULL = nothing masked
    int B, T, L, n_mel;
    float *out, *d_mel, *d_pitch, *d_energy, *d_dur;
    float* partial;                            // [kLossBlocks][6]
    const float* grad_seed;                    // NULL or one float: multiplies every gradient
};
// (kLossBlocks: wavesim_shim.h)
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    const int tid = (int)threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}
static __global__ __launch_bounds__(256) void train_loss_partial_kernel(const LossP p) {
    ESMI_DYN_LDS(red);   // 256 floats
    const long tid = (long)blockIdx.x * 256 + threadIdx.x, nth = (long)kLossBlocks * 256;
    const long nf = (long)p.B * p.L, np_ = (long)p.B * p.T;
    float cnt_f = 0.0f, cnt_p = 0.0f, a = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    for (long r = tid; r < nf; r += nth) cnt_f += (p.mel_mask && p.mel_mask[r]) ? 0.0f : 1.0f;
    const int nm4 = p.n_mel >> 2;
    const long n4 = nf * nm4;
    if ((p.n_mel & 3) == 0 && n4 < 0x7FFFFFFFL && ((reinterpret_cast<uintptr_t>(p.mel_pred) | reinterpret_cast<uintptr_t>(p.mel)) & 15) == 0) {
        // 16 bytes per load, four (prediction, target) pairs in flight per thread, every load unconditional (a clamped index, the
        // surplus masked): the scalar form below ran 94 dependent element loads and as many 64-bit divisions per thread (81 us at
        // B = 128 for 49 MB)
        for (long q0 = tid; q0 < n4; q0 += 4 * nth) {
            f32x4 pv[4], tv[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long qq = q0 + u * nth;
                const unsigned qc = (unsigned)(qq < n4 ? qq : n4 - 1);
                ok[u] = qq < n4 && !(p.mel_mask && p.mel_mask[qc / (unsigned)nm4]);
                pv[u] = ld4(p.mel_pred + (long)qc * 4);
                tv[u] = ld4(p.mel + (long)qc * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!ok[u]) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) a += fabsf(pv[u][e] - tv[u][e]);
            }
        }
    } else {
        for (long q = tid; q < nf * p.n_mel; q += nth)
            if (!(p.mel_mask && p.mel_mask[q / p.n_mel])) a += fabsf(p.mel_pred[q] - p.mel[q]);
    }
    for (long r = tid; r < np_; r += nth) {
        if (p.ph_mask && p.ph_mask[r]) continue;
        const float d1 = p.pitch_pred[r] - p.pitch[r], d2 = p.energy_pred[r] - p.energy[r];
        const float d3 = logf(p.dur_pred[r] + 1.0f) - logf((float)p.dur[r] + 1.0f);
        cnt_p += 1.0f; s1 = fmaf(d1, d1, s1); s2 = fmaf(d2, d2, s2); s3 = fmaf(d3, d3, s3);
    }
    const float v[6] = {cnt_f, cnt_p, a, s1, s2, s3};
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        const float t = block_sum_256(v[e], red);
        if (threadIdx.x == 0) p.partial[(long)blockIdx.x * 6 + e] = t;
    }
}
static __global__ __launch_bounds__(256) void train_loss_final_kernel(const LossP p) {
    ESMI_DYN_LDS(red);
    float t[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) t[e] = block_sum_256((int)threadIdx.x < kLossBlocks ? p.partial[(long)threadIdx.x * 6 + e] : 0.0f, red);
    if (threadIdx.x == 0) {
        const float n_el = t[0] * (float)p.n_mel, n_ph = t[1];
        const float mel_l = t[2] / n_el, l1 = t[3] / n_ph, l2 = t[4] / n_ph, l3 = t[5] / n_ph;
        p.out[0] = mel_l; p.out[1] = l1; p.out[2] = l2; p.out[3] = l3;
        p.out[4] = 10.0f * mel_l + 2.0f * l1 + 2.0f * l2 + l3;
        p.partial[0] = n_el;              // hand the counts to the gradient kernel (stream order)
        p.partial[1] = n_ph;
    }
}
static __global__ void train_loss_grad_kernel(const LossP p) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nm = (long)p.B * p.L * p.n_mel, np_ = (long)p.B * p.T;
    const float n_el = p.partial[0], n_ph = p.partial[1];
    const float seed = p.grad_seed ? p.grad_seed[0] : 1.0f;
    if (q < nm) {
        const bool ok = !(p.mel_mask && p.mel_mask[q / p.n_mel]);
        const float d = p.mel_pred[q] - p.mel[q];
        p.d_mel[q] = ok ? (10.0f * (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) / n_el) * seed : 0.0f;
    }
    if (q < np_) {
        const bool ok = !(p.ph_mask && p.ph_mask[q]);
        const float d1 = p.pitch_pred[q] - p.pitch[q], d2 = p.energy_pred[q] - p.energy[q];
        const float d3 = logf(p.dur_pred[q] + 1.0f) - logf((float)p.dur[q] + 1.0f);
        p.d_pitch[q] = ok ? (2.0f * 2.0f * d1 / n_ph) * seed : 0.0f;
        p.d_energy[q] = ok ? (2.0f * 2.0f * d2 / n_ph) * seed : 0.0f;
        p.d_dur[q] = ok ? (2.0f * d3 / (p.dur_pred[q] + 1.0f) / n_ph) * seed : 0.0f;
    }
}

// ---- AdamW, torch.optim.AdamW's arithmetic (decoupled decay first, bias corrections as two scalars), over a flat buffer.
// Every scalar is derived in double precision (host, or one device thread for the graph variant) and rounded to fp32 once.
struct AdamWScalars { float beta1, one_m_beta1, beta2, one_m_beta2, eps, decay /* 1 - lr wd */, step_size /* lr / bc1 */, bc2_sqrt, gscale; };

__device__ __forceinline__ void adamw_update(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                             float* __restrict__ v, long q, const AdamWScalars& h) {
    const float grad = g[q] * h.gscale;
    float w = p[q] * h.decay;
    const float mm = h.beta1 * m[q] + h.one_m_beta1 * grad;
    const float vv = h.beta2 * v[q] + h.one_m_beta2 * grad * grad;
    m[q] = mm;
    v[q] = vv;
    const float denom = sqrtf(vv) / h.bc2_sqrt + h.eps;
    w -= h.step_size * (mm / denom);
    p[q] = w;
}

static __global__ void train_adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                   long n, AdamWScalars h) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) adamw_update(p, g, m, v, q, h);
}

// the same with the step count and the learning rate read from device memory, so that a captured hipGraph of the whole
// training step replays correctly -- and so that `precision=16`'s dynamic loss scaling needs no host round trip:
//   hyper  = {lr (caller), 1 - lr wd, lr / bc1, sqrt(bc2), skip flag, gradient scale}  (all but [0] written by train_bump_step_kernel)
//   scaler = NULL, or torch.amp.GradScaler's state {scale, growth_factor, backoff_factor, growth_interval, clean steps in a row,
//            skipped steps}: an inf / nan in the (scaled) gradients -- `grad_absmax` holds max|g| with nan ordered as inf -- skips the
//            update (the step counter does not advance), multiplies the scale by backoff_factor; growth_interval clean steps in a
//            row multiply it by growth_factor (GradScaler.step + .update, torch/amp/grad_scaler.py)
static __global__ void train_bump_step_kernel(int* __restrict__ step, float* __restrict__ hyper, double beta1, double beta2, double wd,
                                              const float* __restrict__ grad_absmax, float* __restrict__ scaler) {
    if (threadIdx.x != 0) return;
    bool skip = false;
    float gscale = 1.0f;
    if (scaler) {
        const float amax = grad_absmax[0], scale = scaler[0];
        if (!(amax <= 3.4028234663852886e38f)) {       // inf or nan
            skip = true;
            scaler[0] = scale * scaler[2];
            scaler[4] = 0.0f;
            scaler[5] += 1.0f;
        } else {
            gscale = 1.0f / scale;
            float good = scaler[4] + 1.0f;
            if (good >= scaler[3]) { scaler[0] = scale * scaler[1]; good = 0.0f; }
            scaler[4] = good;
        }
    }
    hyper[4] = skip ? 1.0f : 0.0f;
    hyper[5] = gscale;
    if (skip) return;
    const int t = step[0] + 1;
    step[0] = t;
    const double lr = (double)hyper[0];
    const double bc1 = -expm1((double)t * log(beta1)), bc2 = -expm1((double)t * log(beta2));
    hyper[1] = (float)(1.0 - lr * wd);
    hyper[2] = (float)(lr / bc1);
    hyper[3] = (float)sqrt(bc2);
}
static __global__ void train_adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                       long n, AdamWScalars h, const float* __restrict__ hyper) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n || hyper[4] != 0.0f) return;
    h.decay = hyper[1];
    h.step_size = hyper[2];
    h.bc2_sqrt = hyper[3];
    h.gscale = hyper[5];
    adamw_update(p, g, m, v, q, h);
}

}  // namespace esmi
