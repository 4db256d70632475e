// One AcousticDecoder (pitch, energy or duration predictor) of a dim = 128 model (base ES) over a whole utterance (T <= 256) in one
// workgroup, plus everything of the variance adaptor that hangs on it (layers/networks.py:128-165, :346-384, :233-244):
//     y = ReLU(conv1_k3(fused));  y = ReLU(LN1(y));  y = ReLU(conv2_k3(y));  pred = Linear(dim, 1)(y)            [on the pre-norm2 rows]
//     pitch / energy:  feat[:, (1 + q) dim ...] = Emb(bucketize(pred or target))            duration:  feat[:, 3 dim ...] = mask(LN2(y)),
//     dur = clamp(mask(round(pred or target))), cum = cumsum(dur), mel_len
// grid = (utterances, 3 predictors).  The per-op plan ran this as six GEMM launches with the hidden rows through HBM, a tail kernel and
// the scan (443 us at B = 512, T = 256); enc_va64.h's plan -- activations in registers, LDS for the weights -- carries over when the
// three predictors are separate workgroups: a dim-128 row is 32 registers per tensor and 16-row tile, and a wave holds two tiles of
// {input rows or hidden rows} + the convolution's accumulators + one group of weight fragments = ~210 of its 256 registers.  (All three
// predictors, the Fuse and the head in one workgroup, as for dim 64, would need ~300.)
//   * eight waves x two 16-row tiles; products transposed (lane (i, g): row i, channels 16 nt + 4 g + (0..3) of tile nt);
//   * a convolution = three STEPS, one tap (128 x 128: 64 KB as pre-split fragments) per step, two 64 KB LDS buffers filled by LDS-DMA a
//     step ahead, one workgroup barrier per step; the accumulators stay in registers across the three steps;
//   * row +- 1 operands by DPP row shifts + a 16 KB LDS exchange buffer for the rows across tile boundaries (first and last row of every
//     tile; one buffer serves the fused rows, then the hidden rows);
//   * conv1's result becomes conv2's operand in registers (v_permlane32_swap, enc_va64.h `to_bop`).
// The fused rows come from feat[:, 0 .. dim) (the Fuse stage's launches write them there).
#pragma once
#include "enc_va64.h"

namespace esmi {

struct Pred128Lds {   // floats / dwords
    static constexpr int wbuf = 16 * 1024;                              // one tap: 64 KB
    static constexpr int w0 = 0, w1 = wbuf, par = 2 * wbuf;
    static constexpr int par_sz = 8 * 128;                              // conv1_b, ln1_g, ln1_b, conv2_b, lin_w, ln2_g, ln2_b, edges (127, +inf)
    static constexpr int bnd = par + par_sz, bnd_sz = 16 * 2 * 128;     // [tile][first | last][k group 4][piece 2][16 dwords]
    static constexpr int sdur = bnd + bnd_sz, total = sdur + 256;
};
static_assert(Pred128Lds::total * 4 <= 160 * 1024, "enc_pred128: LDS");
inline int pred128_lds_bytes() { return Pred128Lds::total * (int)sizeof(float); }
enum { PP_C1B = 0, PP_LN1G = 128, PP_LN1B = 256, PP_C2B = 384, PP_LINW = 512, PP_LN2G = 640, PP_LN2B = 768, PP_EDGE = 896 };

namespace p128 {
using namespace c16;
using namespace va64;
constexpr int DIM = 128, KG = 4, NT = 8;
__device__ __forceinline__ void to_bop8(const f32x4 (&v)[NT], f16x2p (&out)[KG], bool lower) {
#pragma unroll
    for (int G = 0; G < KG; ++G) {
        f32x4 recv;
#pragma unroll
        for (int e = 0; e < 4; ++e) recv[e] = swap32_f(lower ? v[2 * G + 1][e] : v[2 * G][e]);
        out[G] = split_f16x2(lower ? v[2 * G] : recv, lower ? recv : v[2 * G + 1]);
    }
}
__device__ __forceinline__ int bnd_at8(int tile, int side, int G, int piece, int g) { return ((tile * 2 + side) * (2 * KG) + G * 2 + piece) * 16 + 4 * g; }
__device__ __forceinline__ void bnd_publish8(unsigned* bnd, int tile0, int i, int g, const f16x2p (&X)[2][KG]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (i == 0 || i == 15) {
            const int side = i == 0 ? 0 : 1;
#pragma unroll
            for (int G = 0; G < KG; ++G) {
                *reinterpret_cast<u32x4*>(bnd + bnd_at8(tile0 + t, side, G, 0, g)) = X[t][G].h1;
                *reinterpret_cast<u32x4*>(bnd + bnd_at8(tile0 + t, side, G, 1, g)) = X[t][G].h2;
            }
        }
    }
}
__device__ __forceinline__ f16x2p bnd_read8(const unsigned* bnd, int tile, int side, int G, int g, bool exists) {
    f16x2p o = zero_bop();
    if (exists) {
        o.h1 = *reinterpret_cast<const u32x4*>(bnd + bnd_at8(tile, side, G, 0, g));
        o.h2 = *reinterpret_cast<const u32x4*>(bnd + bnd_at8(tile, side, G, 1, g));
    }
    return o;
}
// one tap of a k = 3 convolution: c[t][nt] += W_j . X^T(row + j - 1) over the four k groups; W = the tap's 64 KB in LDS
__device__ __forceinline__ void conv_tap(f32x4 (&c)[2][NT], const float* W, int lw, int j, const f16x2p (&X)[2][KG], const unsigned* bnd,
                                         int tile0, int ntiles, int g) {
#pragma unroll
    for (int G = 0; G < KG; ++G) {
        f16x2p op[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int tile = tile0 + t;
            if (j == 0) op[t] = rows_dn(X[t][G], bnd_read8(bnd, tile - 1, 1, G, g, tile > 0));
            else if (j == 1) op[t] = X[t][G];
            else op[t] = rows_up(X[t][G], bnd_read8(bnd, tile + 1, 0, G, g, tile + 1 < ntiles));
        }
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {      // four 16-channel output tiles at a time (the fragments of all eight are 64 registers)
            WFrags<4> wf;
            wfrags_load<4, 4, 4>(wf, 0, W + 2 * ch * 256, lw, G);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 (&acc4)[4] = *reinterpret_cast<f32x4 (*)[4]>(&c[t][4 * ch]);
                mma_all<4>(acc4, wf, op[t]);
            }
        }
    }
}
}  // namespace p128

__global__ __launch_bounds__(64 * 8, 1) void enc_pred128_kernel(const Pred128P p) {
    using namespace c16;
    using namespace va64;
    using namespace p128;
    typedef Pred128Lds M;
    ESMI_DYN_LDS(lds);
    const int nw = (int)(blockDim.x >> 6), w = uniform_i(wave_id());
    const int lane = lane_id(), i = lane & 15, g = lane >> 4;
    const bool lower = lane < 32;
    const int b = (int)blockIdx.x, q = (int)blockIdx.y;       // utterance, predictor (0 pitch, 1 energy, 2 duration)
    const int tile0 = 2 * w, ntiles = 2 * nw, rot = (int)(blockIdx.x + blockIdx.y);
    float* const wb[2] = {lds + M::w0, lds + M::w1};
    float* const par = lds + M::par;
    unsigned* const bnd = reinterpret_cast<unsigned*>(lds) + M::bnd;
    int* const sdur = reinterpret_cast<int*>(lds + M::sdur);
    const int lw = wlane(lane, 4);
    const f32x4 z4 = zero4();
    const PredW& d = p.pred[q];
    // step k = 0..5: tap k % 3 of conv1 (k < 3) / conv2, in buffer k & 1
    auto request = [&](int k) __attribute__((always_inline)) {
        dma_frags((k < 3 ? d.conv1_w : d.conv2_w) + (k % 3) * (64 * 256), wb[k & 1], 64, w, nw, lane, rot);
    };
    auto step_begin = [&](int k) __attribute__((always_inline)) {
        wait_vm0();
        wg_sync_lds();
        if (k >= 1 && k + 1 < 6) request(k + 1);
    };
    request(0);
    request(1);
    {   // parameter vectors: 128 floats = half an instruction each (lanes 32.. copy the next vector)
        const int v2 = lane >> 5, c4 = 4 * (lane & 31);
        if (w == 0 % nw) lds_dma16((v2 ? d.ln1_g : d.conv1_b) + c4, par + PP_C1B, lane);
        if (w == 1 % nw) lds_dma16((v2 ? d.conv2_b : d.ln1_b) + c4, par + PP_LN1B, lane);
        if (w == 2 % nw) lds_dma16((v2 ? d.ln2_g : d.lin_w) + c4, par + PP_LINW, lane);
        if (w == 3 % nw) lds_dma16((v2 ? d.ln2_b : d.ln2_b) + c4, par + PP_LN2B, lane);     // (the upper half lands in the edge slots: overwritten below)
    }
    const float lin_b = d.lin_b[0];
    const BufRsrc r_mask = make_rsrc(p.mask ? p.mask + (long)b * p.T : nullptr, p.T);
    const BufRsrc r_feat = make_rsrc(p.feat + (long)b * p.T * 4 * DIM, (long)p.T * 4 * DIM * 4);
    int pos[2];
    bool rout[2], rz[2];
    f16x2p X[2][KG];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        pos[t] = 16 * (tile0 + t) + i;
        rout[t] = pos[t] >= p.T;
        rz[t] = !rout[t] && buf_ld_u8(r_mask, (unsigned)pos[t]) != 0;
        const unsigned o0 = rout[t] ? kBufOOB : (unsigned)(pos[t] * 4 * DIM * 4) + gl_lane(lane);
#pragma unroll
        for (int G = 0; G < KG; ++G) X[t][G] = global_bop(r_feat, o0, G);
    }
    bnd_publish8(bnd, tile0, i, g, X);
    f32x4 c[2][NT];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) c[t][nt] = z4;
    }
    // ---------------- conv1 (three taps), ReLU, LayerNorm 1, ReLU -> the hidden rows as conv2's operand
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        step_begin(j);
        if (j == 0 && q < 2 && w == 0) {   // bucket edges (dim - 1 of them), +inf behind: two floats per lane
            const float* bins = d.bins;
            par[PP_EDGE + lane] = bins[lane];
            par[PP_EDGE + 64 + lane] = lane < 63 ? bins[64 + lane] : INFINITY;
        }
        conv_tap(c, wb[j & 1], lw, j, X, bnd, tile0, ntiles, g);
    }
    f16x2p H[2][KG];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x4 gg[NT], bb[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            c[t][nt] = relu4(fmaf4(c[t][nt], kF16WScaleInv, ld4_lds(par + PP_C1B + 16 * nt + 4 * g)));
            gg[nt] = ld4_lds(par + PP_LN1G + 16 * nt + 4 * g);
            bb[nt] = ld4_lds(par + PP_LN1B + 16 * nt + 4 * g);
        }
        layernorm<NT>(c[t], gg, bb);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) c[t][nt] = rout[t] ? z4 : relu4(c[t][nt]);
        to_bop8(c[t], H[t], lower);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) c[t][nt] = z4;
    }
    wg_sync_lds();                 // every wave is through its last read of the fused rows' boundary rows: the buffer takes the hidden rows'
    bnd_publish8(bnd, tile0, i, g, H);
    // ---------------- conv2 (three taps), ReLU, Linear(dim, 1) on the pre-norm2 rows
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        step_begin(3 + j);
        conv_tap(c, wb[(3 + j) & 1], lw, j, H, bnd, tile0, ntiles, g);
    }
    const BufRsrc r_pred = make_rsrc(p.preds[q] + (long)b * p.T, (long)p.T * 4);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float s = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            c[t][nt] = relu4(fmaf4(c[t][nt], kF16WScaleInv, ld4_lds(par + PP_C2B + 16 * nt + 4 * g)));
            const f32x4 lwv = ld4_lds(par + PP_LINW + 16 * nt + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) s = fmaf(c[t][nt][e], lwv[e], s);
        }
        float pr = row_sum4(s) + lin_b;
        if (q == 2) pr = fmaxf(pr, 0.0f);
        const unsigned frow = rout[t] ? kBufOOB : (unsigned)(pos[t] * 4 * DIM * 4);
        const unsigned srow = (!rout[t] && g == 0) ? (unsigned)(pos[t] * 4) : kBufOOB;   // one lane per row
        buf_st(r_pred, srow, pr);
        if (q < 2) {   // torch.bucketize(v, edges, right=False) = number of edges strictly below v; the embedding row -> feat
            const float* tv = q == 0 ? p.pitch_t : p.energy_t;
            const float v = (tv && !rout[t]) ? tv[(long)b * p.T + pos[t]] : pr;
            float cnt = 0.0f;
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const f32x4 e0 = ld4_lds(par + PP_EDGE + 32 * g + 4 * k4);
#pragma unroll
                for (int e = 0; e < 4; ++e) cnt += e0[e] < v ? 1.0f : 0.0f;
            }
            const int bidx = (int)row_sum4(cnt);
            int* idx = q == 0 ? p.pitch_idx : p.energy_idx;
            if (idx) {
                const BufRsrc r_i = make_rsrc(idx + (long)b * p.T, (long)p.T * 4);
                buf_st_i(r_i, srow, bidx);
            }
            const float* row = d.emb + bidx * DIM + 4 * g;             // the row's four lanes copy 64 contiguous bytes per instruction
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4)
                buf_st4(r_feat, frow == kBufOOB ? kBufOOB : frow + (unsigned)(((1 + q) * DIM + 16 * k4 + 4 * g) * 4), rz[t] ? z4 : ld4(row + 16 * k4));
        } else {       // duration features (networks.py:161-163) = mask(LN2(y)); rounded durations; the length regulator's scan
            f32x4 gg[NT], bb[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                gg[nt] = ld4_lds(par + PP_LN2G + 16 * nt + 4 * g);
                bb[nt] = ld4_lds(par + PP_LN2B + 16 * nt + 4 * g);
            }
            layernorm<NT>(c[t], gg, bb);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                buf_st4(r_feat, frow == kBufOOB ? kBufOOB : frow + (unsigned)((3 * DIM + 16 * nt + 4 * g) * 4), rz[t] ? z4 : c[t][nt]);
            float dval = (p.dur_t && !rout[t]) ? (float)p.dur_t[(long)b * p.T + pos[t]] : rintf(pr);   // torch.round: half to even
            if (p.mask) {                                                                                 // networks.py:381-382
                if (rz[t]) dval = 0.0f;
                dval = fmaxf(dval, 0.0f);
            }
            if (g == 0) sdur[pos[t]] = rout[t] ? 0 : max((int)dval, 0);
            const BufRsrc r_dur = make_rsrc(p.dur + (long)b * p.T, (long)p.T * 4);
            buf_st_i(r_dur, srow, (int)dval);
        }
    }
    if (q == 2 && p.cum) {   // FeatureUpsampler's scan (networks.py:233-244); T <= 256
        wg_sync_lds();
        if (w == 0) {
            const int per = (p.T + 63) / 64, q0 = lane * per;
            int local = 0;
            for (int e = 0; e < per; ++e) local += (q0 + e < p.T) ? sdur[q0 + e] : 0;
            int incl = local;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                const int v = shfl_up_i(incl, dd);
                if (lane >= dd) incl += v;
            }
            const BufRsrc r_cum = make_rsrc(p.cum + (long)b * p.T, (long)p.T * 4);
            int run = incl - local;
            for (int e = 0; e < per; ++e) {
                run += (q0 + e < p.T) ? sdur[q0 + e] : 0;
                buf_st_i(r_cum, (q0 + e < p.T) ? (unsigned)((q0 + e) * 4) : kBufOOB, run);
            }
            const int total = shfl_i(incl, 63);
            if (lane == 0) p.mel_len[b] = total;
        }
    }
}

}  // namespace esmi
