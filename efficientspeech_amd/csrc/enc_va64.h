// Fuse + variance adaptor (+ the length regulator's scan) for dim = 64 models (small ES) whose sequence one workgroup covers
// (T <= 256, two encoder levels, ConvTranspose kernel 3): round 6.  Same reference operations as enc_fuse_va.h / enc_va16.h
// (layers/networks.py:189-219, :128-165, :346-384, :233-244) and, when the caller wants it, the decoder's phoneme-rate first stage
// h0 = LN(tanh(feat Wp^T + b)) (the row-wise head of :291-294, 4 dim = dx2 = 256) as eight more weight steps per tile behind the
// predictors -- its four operands (fused rows, the two embedding rows, the duration features) are on the CU at that moment, so the
// (B, T, 256) feature tensor is neither written nor read back when nobody else wants it.
//
// Why not enc_va16.h with DIM = 64: its plan keeps the fused rows and the three predictors' hidden rows as shared LDS tiles
// (dim 64, 256 rows: 70 KB + 210 KB) and the weights in two 36 KB halves (one dim-64 k = 3 convolution alone is 48 KB).  Here NO
// activation tile lives in LDS:
//   * a wave owns NTILE (1 or 2) 16-row tiles from the input rows to the stored features; products are transposed (lane = row:
//     lane (i, g) holds row i and, per 16-channel tile nt, channels 16 nt + 4 g + (0..3));
//   * a GEMM's output becomes the next GEMM's second operand IN REGISTERS: that operand wants channels {c0..c0+3, c0+8..c0+11},
//     c0 = 32 G + 16 (g >> 1) + 4 (g & 1), of the lane's row -- the lane's own four channels of tile 2 G (g < 2) or 2 G + 1 (g >= 2)
//     plus its partner's (lane ^ 32): one v_permlane32_swap per register (`to_bop`), then the f16 split;
//   * the k = 3 convolutions take their row +-1 operands by DPP row shifts (v_mov_dpp row_shr:1 / row_shl:1 inside the 16 lanes that
//     hold a tile's rows for one g); the rows across a tile boundary come from a 4 KB LDS exchange buffer (first and last row of
//     every tile), written at the end of the producing step -- the only activation bytes in LDS;
//   * LDS therefore belongs to the WEIGHTS: two 48 KB buffers (one k = 3 convolution of one predictor, or a Fuse stage), filled by
//     LDS-DMA one step ahead; nine steps (mlp 0 + mlp 1 | ConvTranspose | fuse Linear | conv1, conv2 of pitch, energy, duration), one
//     workgroup barrier per step (it publishes the step's weights, releases the other buffer and orders the boundary rows);
//   * the level-1 row a position needs from the ConvTranspose (n = pos >> 1) is computed by the position's own lane (Linear is
//     row-wise: every level-1 row is computed twice), row n - 1 of an even position is the row above's: the Fuse needs no gather
//     through LDS; taps that do not apply to a row get a zero operand.
// Weights: the arrays esmi_pack_bfrag_f32 makes (chain16.h header: NTW = 2 tiles of 32 rows, 16 KiB per 64 x 64 matrix or tap).
#pragma once
#include "chain16.h"
#include "enc_fuse_va.h"

namespace esmi {

constexpr int kVa64MaxWaves = 8;
constexpr int kVa64Dim = 64;

struct Va64Lds {   // floats / dwords
    static constexpr int wbuf = 12 * 1024;                              // one weight buffer: 48 KB
    static constexpr int w0 = 0, w1 = wbuf;
    static constexpr int par = 2 * wbuf;                                // parameter vectors, see VP_*
    static constexpr int par_sz = 512 + 3 * 512 + 768;                  // (+ the decoder head's bias | gain | shift)
    static constexpr int bnd_sz = 2 * kVa64MaxWaves * 2 * 64;           // [tile][first | last][k group 2][piece 2][16 dwords]
    static constexpr int bndF = par + par_sz;
    static constexpr int bndH = bndF + bnd_sz;
    static constexpr int sdur = bndH + bnd_sz;                          // [256] ints
    static constexpr int total = sdur + 256;
};
static_assert(Va64Lds::total * 4 <= 160 * 1024, "enc_va64: LDS");
inline int va64_lds_bytes() { return Va64Lds::total * (int)sizeof(float); }
// parameter vectors (float offsets inside Va64Lds::par); four 64-float vectors per LDS-DMA instruction
enum { VP_MLPB0 = 0, VP_MLPB1 = 64, VP_UPB1 = 128, VP_FUSEB = 192, VP_LN2G = 256, VP_LN2B = 320, VP_EDGE = 384 /* pitch, energy: 63 edges, +inf */,
       VP_PRED = 512 /* + 512 q: conv1_b, ln1_g, ln1_b, conv2_b | lin_w, 3 unused */, VP_HEADB = 2048, VP_HEADG = 2304, VP_HEADBE = 2560 };

namespace va64 {
using namespace c16;

// rows one down / one up inside the 16 lanes that hold a tile's rows for one g (row_dn_u / row_up_u, wavesim_shim.h): the lane at the
// tile's edge takes `edge` (the neighbouring tile's row, or zero outside the sequence)
__device__ __forceinline__ f16x2p rows_dn(const f16x2p& x, const f16x2p& edge) {
    f16x2p o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o.h1[e] = row_dn_u(x.h1[e], edge.h1[e]); o.h2[e] = row_dn_u(x.h2[e], edge.h2[e]); }
    return o;
}
__device__ __forceinline__ f16x2p rows_up(const f16x2p& x, const f16x2p& edge) {
    f16x2p o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o.h1[e] = row_up_u(x.h1[e], edge.h1[e]); o.h2[e] = row_up_u(x.h2[e], edge.h2[e]); }
    return o;
}
__device__ __forceinline__ f16x2p zero_bop() {
    f16x2p o;
    o.h1 = u32x4{0u, 0u, 0u, 0u};
    o.h2 = u32x4{0u, 0u, 0u, 0u};
    return o;
}
// D^T rows (4 tiles of 16 channels) -> the second operand of the next GEMM's two k groups (see the header)
__device__ __forceinline__ void to_bop(const f32x4 (&v)[4], f16x2p (&out)[2], bool lower) {
#pragma unroll
    for (int G = 0; G < 2; ++G) {
        f32x4 recv;
#pragma unroll
        for (int e = 0; e < 4; ++e) recv[e] = swap32_f(lower ? v[2 * G + 1][e] : v[2 * G][e]);
        out[G] = split_f16x2(lower ? v[2 * G] : recv, lower ? recv : v[2 * G + 1]);
    }
}
// boundary rows: dword index of (tile, side, k group, piece) for lane group g
__device__ __forceinline__ int bnd_at(int tile, int side, int G, int piece, int g) { return ((tile * 2 + side) * 4 + G * 2 + piece) * 16 + 4 * g; }
template <int NTILE>
__device__ __forceinline__ void bnd_publish(unsigned* bnd, int tile0, int i, int g, const f16x2p (&X)[NTILE][2]) {
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        if (i == 0 || i == 15) {
            const int side = i == 0 ? 0 : 1;
#pragma unroll
            for (int G = 0; G < 2; ++G) {
                *reinterpret_cast<u32x4*>(bnd + bnd_at(tile0 + t, side, G, 0, g)) = X[t][G].h1;
                *reinterpret_cast<u32x4*>(bnd + bnd_at(tile0 + t, side, G, 1, g)) = X[t][G].h2;
            }
        }
    }
}
__device__ __forceinline__ f16x2p bnd_read(const unsigned* bnd, int tile, int side, int G, int g, bool exists) {
    f16x2p o = zero_bop();
    if (exists) {
        o.h1 = *reinterpret_cast<const u32x4*>(bnd + bnd_at(tile, side, G, 0, g));
        o.h2 = *reinterpret_cast<const u32x4*>(bnd + bnd_at(tile, side, G, 1, g));
    }
    return o;
}

// c[t][nt] += sum over the three taps and the two k groups of W_j . X^T(row + j - 1), W = one k = 3 convolution (48 KB) in LDS;
// every weight fragment is read once for all NTILE tiles of the wave
template <int NTILE>
__device__ __forceinline__ void conv3(f32x4 (&c)[NTILE][4], const float* W, int lw, const f16x2p (&X)[NTILE][2], const unsigned* bnd, int tile0,
                                      int ntiles, int g) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        f16x2p op[NTILE][2];        // the tap's operand rows (shifted one tap at a time: both shifts of both tiles at once are 64 registers)
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
            const int tile = tile0 + t;
#pragma unroll
            for (int G = 0; G < 2; ++G) {
                if (j == 0) op[t][G] = rows_dn(X[t][G], bnd_read(bnd, tile - 1, 1, G, g, tile > 0));
                else if (j == 1) op[t][G] = X[t][G];
                else op[t][G] = rows_up(X[t][G], bnd_read(bnd, tile + 1, 0, G, g, tile + 1 < ntiles));
            }
        }
#pragma unroll
        for (int G = 0; G < 2; ++G) {
            WFrags<4> wf;
            wfrags_load<4, 2, 4>(wf, 0, W + j * (16 * 256), lw, G);
#pragma unroll
            for (int t = 0; t < NTILE; ++t) mma_all<4>(c[t], wf, op[t][G]);
        }
    }
}
// acc[t][nt] += W[.., 32 (G0 + ks) ..] . X[t][ks]^T for ks < KS: W a packed matrix of NTW = 2 row tiles in LDS
template <int NTILE, int KS>
__device__ __forceinline__ void gemm_tiles(f32x4 (&acc)[NTILE][4], const float* W, int lw, int G0, const f16x2p (&X)[NTILE][KS]) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        WFrags<4> wf;
        wfrags_load<4, 2, 4>(wf, 0, W, lw, G0 + ks);
#pragma unroll
        for (int t = 0; t < NTILE; ++t) mma_all<4>(acc[t], wf, X[t][ks]);
    }
}
}  // namespace va64

template <int NTILE>
__device__ __forceinline__ void enc_va64_body(const FuseVaP& p) {
    using namespace c16;
    using namespace va64;
    typedef Va64Lds M;
    constexpr int DIM = kVa64Dim;
    ESMI_DYN_LDS(lds);
    const int nw = (int)(blockDim.x >> 6), w = uniform_i(wave_id());
    const int lane = lane_id(), i = lane & 15, g = lane >> 4;
    const bool lower = lane < 32;
    const int b = (int)blockIdx.x;
    const int tile0 = NTILE * w, ntiles = NTILE * nw;
    float* const wb[2] = {lds + M::w0, lds + M::w1};
    float* const par = lds + M::par;
    unsigned* const bndF = reinterpret_cast<unsigned*>(lds) + M::bndF;
    unsigned* const bndH = reinterpret_cast<unsigned*>(lds) + M::bndH;
    int* const sdur = reinterpret_cast<int*>(lds + M::sdur);
    const int lw = wlane(lane, 2);
    const f32x4 z4 = zero4();
    const int rot = (int)blockIdx.x;
    const int n1 = p.n_i[1];

    // ---------------- the nine weight sets (48 KB buffers, LDS-DMA): request(k) copies set k into buffer k & 1
    auto request = [&](int k) __attribute__((always_inline)) {
        float* dst = wb[k & 1];
        if (k == 0) {
            dma_frags(p.mlp_w[0], dst, 16, w, nw, lane, rot);
            dma_frags(p.mlp_w[1], dst + 16 * 256, 32, w, nw, lane, rot);
        } else if (k == 1) {
            dma_frags(p.up_w[1], dst, 48, w, nw, lane, rot);
        } else if (k == 2) {
            dma_frags(p.fuse_w, dst, 32, w, nw, lane, rot);
        } else if (k < 9) {
            const int q = (k - 3) >> 1;
            dma_frags((k - 3) & 1 ? p.pred[q].conv2_w : p.pred[q].conv1_w, dst, 48, w, nw, lane, rot);
        } else {   // the head's k group (k - 9) & 7: 32 channels x 256 outputs = 32 KB (streamed once per tile of the wave)
            dma_frags(p.head_w + ((k - 9) & 7) * (32 * 256), dst, 32, w, nw, lane, rot);
        }
    };
    const int nset = p.h0 ? 9 + 8 * NTILE : 9;
    // step k begins: this wave's share of set k has landed, every wave is through step k - 1 (the other buffer is free, the boundary
    // rows written in step k - 1 are visible); then set k + 1 is requested into the buffer step k - 1 used
    auto step_begin = [&](int k) __attribute__((always_inline)) {
        wait_vm0();
        wg_sync_lds();
        if (k >= 1 && k + 1 < nset) request(k + 1);
    };

    // ---------------- entry: the rows' own inputs, the first two weight sets and every parameter vector on their way
    request(0);
    request(1);
    {
        const int v4 = lane >> 4, c4 = 4 * (lane & 15);
        auto pick4 = [&](const float* a0, const float* a1, const float* a2, const float* a3) __attribute__((always_inline)) {
            return (v4 & 2 ? (v4 & 1 ? a3 : a2) : (v4 & 1 ? a1 : a0)) + c4;
        };
        if (w == 0 % nw) lds_dma16(pick4(p.mlp_b[0], p.mlp_b[1], p.up_b[1], p.fuse_b), par + VP_MLPB0, lane);
        if (w == 1 % nw) lds_dma16(pick4(p.pred[2].ln2_g, p.pred[2].ln2_b, p.fuse_b, p.fuse_b), par + VP_LN2G, lane);   // (the edge slots are written below)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const PredW& d = p.pred[q];
            if (w == (2 + 2 * q) % nw) lds_dma16(pick4(d.conv1_b, d.ln1_g, d.ln1_b, d.conv2_b), par + VP_PRED + 512 * q, lane);
            if (w == (3 + 2 * q) % nw) lds_dma16(pick4(d.lin_w, d.lin_w, d.lin_w, d.lin_w), par + VP_PRED + 512 * q + 256, lane);
        }
    }
    if (p.h0) {   // head bias | gain | shift: 256 floats = one instruction each
        if (w == 0 % nw) lds_dma16(p.head_b + 4 * lane, par + VP_HEADB, lane);
        if (w == 1 % nw) lds_dma16(p.head_g + 4 * lane, par + VP_HEADG, lane);
        if (w == 2 % nw) lds_dma16(p.head_beta + 4 * lane, par + VP_HEADBE, lane);
    }
    const float lb0 = p.pred[0].lin_b[0], lb1 = p.pred[1].lin_b[0], lb2 = p.pred[2].lin_b[0];
    const BufRsrc r_mask = make_rsrc(p.mask ? p.mask + (long)b * p.T : nullptr, p.T);
    const BufRsrc r_feat = make_rsrc(p.feat ? p.feat + (long)b * p.T * 4 * DIM : nullptr, (long)p.T * 4 * DIM * 4);
    const BufRsrc r_pt = make_rsrc(p.pitch_t ? p.pitch_t + (long)b * p.T : nullptr, (long)p.T * 4);
    const BufRsrc r_et = make_rsrc(p.energy_t ? p.energy_t + (long)b * p.T : nullptr, (long)p.T * 4);
    const BufRsrc r_dt = make_rsrc(p.dur_t ? p.dur_t + (long)b * p.T : nullptr, (long)p.T * 4);
    const BufRsrc r_f0 = make_rsrc(p.feats[0] + (long)b * p.n_i[0] * DIM, (long)p.n_i[0] * DIM * 4);
    const BufRsrc r_f1 = make_rsrc(p.feats[1] + (long)b * n1 * 2 * DIM, (long)n1 * 2 * DIM * 4);
    int pos[NTILE];
    bool rout[NTILE], rz[NTILE];
    float tv_p[NTILE], tv_e[NTILE], tv_d[NTILE];
    f16x2p X0[NTILE][2], Xa[NTILE][4];      // second GEMM operands: the level-0 row; the level-1 row n = pos >> 1
    bool na_ok[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        pos[t] = 16 * (tile0 + t) + i;
        rout[t] = pos[t] >= p.T;
        rz[t] = !rout[t] && buf_ld_u8(r_mask, (unsigned)pos[t]) != 0;
        tv_p[t] = tv_e[t] = tv_d[t] = 0.0f;      // (the teacher values are requested two steps before their use, below: nine steps of life
                                                 // for six registers is what the two-tile instantiation spilled)
        const unsigned o0 = rout[t] ? kBufOOB : (unsigned)(pos[t] * DIM * 4) + gl_lane(lane);
        const int na = pos[t] >> 1;
        na_ok[t] = !rout[t] && na < n1;
        const unsigned oa = na_ok[t] ? (unsigned)(na * 2 * DIM * 4) + gl_lane(lane) : kBufOOB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) X0[t][ks] = global_bop(r_f0, o0, ks);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) Xa[t][ks] = global_bop(r_f1, oa, ks);
    }

    // ================================================================ step 0: Linear of level 0; Linear of level 1 on row n = pos >> 1 (every level-1 row
    // is computed by the two positions it feeds: the transposed convolution below then needs no gather)
    step_begin(0);
    f16x2p C0[NTILE][2], Ya[NTILE][2];
    {
        f32x4 a0[NTILE][4], aa[NTILE][4];
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { a0[t][nt] = z4; aa[t][nt] = z4; }
        }
        gemm_tiles<NTILE, 2>(a0, wb[0], lw, 0, X0);
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
            f32x4 v0[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) v0[nt] = fmaf4(a0[t][nt], kF16WScaleInv, ld4_lds(par + VP_MLPB0 + 16 * nt + 4 * g));
            to_bop(v0, C0[t], lower);
        }
        sched_fence();    // (level 0 is through before level 1 starts: both at once is what the two-tile instantiation cannot hold)
        gemm_tiles<NTILE, 4>(aa, wb[0] + 16 * 256, lw, 0, Xa);
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
            f32x4 va[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                va[nt] = na_ok[t] ? fmaf4(aa[t][nt], kF16WScaleInv, ld4_lds(par + VP_MLPB1 + 16 * nt + 4 * g)) : z4;   // rows that do not exist contribute nothing
            to_bop(va, Ya[t], lower);
        }
        bnd_publish<NTILE>(bndH, tile0, i, g, Ya);    // (the hidden rows' exchange buffer is free until step 3)
    }
    // ================================================================ step 1: ConvTranspose1d(stride 2, k = 3), cropped to T:
    // out[pos] = W_0 y1[pos/2] + W_2 y1[pos/2 - 1] (pos even) | W_1 y1[(pos-1)/2] (pos odd); a tap that does not apply gets a zero operand.
    // y1[pos/2 - 1] of an even position is what the position above it computed (its n is (pos - 1) >> 1): one row shift.
    const int e_i = lane < DIM - 1 ? lane : DIM - 2;
    const float edge_p = p.pred[0].bins[e_i], edge_e = p.pred[1].bins[e_i];      // bucket edges (dim - 1 of them), requested across the barrier
    step_begin(1);
    if (w == 0) {   // ... and written with +inf behind them (read seven barriers later)
        par[VP_EDGE + lane] = lane < DIM - 1 ? edge_p : INFINITY;
        par[VP_EDGE + 64 + lane] = lane < DIM - 1 ? edge_e : INFINITY;
    }
    f16x2p C1[NTILE][2];
    {
        f32x4 u[NTILE][4];
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) u[t][nt] = z4;
        }
        const f16x2p zb = zero_bop();
        f16x2p Yb[NTILE][2];
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
#pragma unroll
            for (int G = 0; G < 2; ++G) Yb[t][G] = rows_dn(Ya[t][G], bnd_read(bndH, tile0 + t - 1, 1, G, g, tile0 + t > 0));
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int G = 0; G < 2; ++G) {
                WFrags<4> wf;
                wfrags_load<4, 2, 4>(wf, 0, wb[1] + j * (16 * 256), lw, G);
#pragma unroll
                for (int t = 0; t < NTILE; ++t) {
                    const bool even = (pos[t] & 1) == 0;
                    const f16x2p op = j == 0 ? (even ? Ya[t][G] : zb) : (j == 1 ? (even ? zb : Ya[t][G]) : (even ? Yb[t][G] : zb));
                    mma_all<4>(u[t], wf, op);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
            f32x4 v[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) v[nt] = fmaf4(u[t][nt], kF16WScaleInv, ld4_lds(par + VP_UPB1 + 16 * nt + 4 * g));
            to_bop(v, C1[t], lower);
        }
    }
    // ================================================================ step 2: Linear(2 dim, dim) on the concatenation, masked_fill
    step_begin(2);
    f16x2p F[NTILE][2];
    {
        f32x4 a[NTILE][4];
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) a[t][nt] = z4;
        }
        gemm_tiles<NTILE, 2>(a, wb[0], lw, 0, C0);
        gemm_tiles<NTILE, 2>(a, wb[0], lw, 2, C1);
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
            f32x4 fz[4];
            const unsigned frow = rout[t] ? kBufOOB : (unsigned)(pos[t] * 4 * DIM * 4);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                fz[nt] = (rout[t] || rz[t]) ? z4 : fmaf4(a[t][nt], kF16WScaleInv, ld4_lds(par + VP_FUSEB + 16 * nt + 4 * g));   // outside rows = the convs' zero padding
                if (p.feat) buf_st4(r_feat, frow + (unsigned)((16 * nt + 4 * g) * 4), fz[nt]);
            }
            to_bop(fz, F[t], lower);
        }
        bnd_publish<NTILE>(bndF, tile0, i, g, F);
    }
    // ================================================================ steps 3..8: per predictor conv1 (k = 3) -> ReLU -> LayerNorm -> ReLU, then
    // conv2 (k = 3) -> ReLU -> Linear(dim, 1) on the pre-norm2 rows
    float pr[3][NTILE];
    f32x4 cdur[NTILE][4];          // the duration predictor's pre-norm2 rows (its LayerNorm output is a quarter of the feature row)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float* pv = par + VP_PRED + 512 * q;
        f16x2p H[NTILE][2];
        if (q == 2) {
#pragma unroll
            for (int t = 0; t < NTILE; ++t) {
                const unsigned trow = rout[t] ? kBufOOB : (unsigned)(pos[t] * 4);
                tv_p[t] = buf_ld(r_pt, trow); tv_e[t] = buf_ld(r_et, trow); tv_d[t] = buf_ld(r_dt, trow);
            }
        }
        step_begin(3 + 2 * q);
        {
            f32x4 c[NTILE][4];
#pragma unroll
            for (int t = 0; t < NTILE; ++t) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) c[t][nt] = z4;
            }
            conv3<NTILE>(c, wb[(3 + 2 * q) & 1], lw, F, bndF, tile0, ntiles, g);
#pragma unroll
            for (int t = 0; t < NTILE; ++t) {
                f32x4 v[4], gg[4], bb[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    v[nt] = relu4(fmaf4(c[t][nt], kF16WScaleInv, ld4_lds(pv + 16 * nt + 4 * g)));
                    gg[nt] = ld4_lds(pv + 64 + 16 * nt + 4 * g);
                    bb[nt] = ld4_lds(pv + 128 + 16 * nt + 4 * g);
                }
                layernorm<4>(v, gg, bb);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) v[nt] = rout[t] ? z4 : relu4(v[nt]);
                to_bop(v, H[t], lower);
            }
            bnd_publish<NTILE>(bndH, tile0, i, g, H);
        }
        step_begin(4 + 2 * q);
        {
            f32x4 c[NTILE][4];
#pragma unroll
            for (int t = 0; t < NTILE; ++t) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) c[t][nt] = z4;
            }
            conv3<NTILE>(c, wb[(4 + 2 * q) & 1], lw, H, bndH, tile0, ntiles, g);
#pragma unroll
            for (int t = 0; t < NTILE; ++t) {
                float s = 0.0f;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    c[t][nt] = relu4(fmaf4(c[t][nt], kF16WScaleInv, ld4_lds(pv + 192 + 16 * nt + 4 * g)));
                    const f32x4 lwv = ld4_lds(pv + 256 + 16 * nt + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) s = fmaf(c[t][nt][e], lwv[e], s);
                    if (q == 2) cdur[t][nt] = c[t][nt];
                }
                pr[q][t] = row_sum4(s) + (q == 0 ? lb0 : (q == 1 ? lb1 : lb2));
            }
        }
    }
    // ================================================================ bucketize, embeddings, duration features, durations, outputs
    int hidx[NTILE][2];            // bucket indices and duration-feature operands for the head
    f16x2p DF[NTILE][2];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        pr[2][t] = fmaxf(pr[2][t], 0.0f);
        const unsigned frow = rout[t] ? kBufOOB : (unsigned)(pos[t] * 4 * DIM * 4);
        int bidx[2];                // torch.bucketize(v, edges, right=False) = number of edges strictly below v
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool has_t = q == 0 ? p.pitch_t != nullptr : p.energy_t != nullptr;
            const float v = (has_t && !rout[t]) ? (q == 0 ? tv_p[t] : tv_e[t]) : pr[q][t];
            float cnt = 0.0f;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const f32x4 e0 = ld4_lds(par + VP_EDGE + 64 * q + 16 * g + 4 * k4);
#pragma unroll
                for (int e = 0; e < 4; ++e) cnt += e0[e] < v ? 1.0f : 0.0f;
            }
            bidx[q] = (int)row_sum4(cnt);
            hidx[t][q] = bidx[q];
            if (p.feat) {           // the embedding row: lane group g copies floats [16 g, 16 g + 16) of it
                const float* row = (q == 0 ? p.pred[0].emb : p.pred[1].emb) + bidx[q] * DIM + 16 * g;
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4)
                    buf_st4(r_feat, frow + (unsigned)(((1 + q) * DIM + 16 * g + 4 * k4) * 4), rz[t] ? z4 : ld4(row + 4 * k4));
            }
        }
        {   // duration features (networks.py:161-163): LayerNorm 2 of the duration predictor, masked
            f32x4 gg[4], bb[4], df[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                gg[nt] = ld4_lds(par + VP_LN2G + 16 * nt + 4 * g);
                bb[nt] = ld4_lds(par + VP_LN2B + 16 * nt + 4 * g);
                df[nt] = cdur[t][nt];
            }
            layernorm<4>(df, gg, bb);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                if (rz[t]) df[nt] = z4;
                if (p.feat) buf_st4(r_feat, frow + (unsigned)((3 * DIM + 16 * nt + 4 * g) * 4), df[nt]);
            }
            if (p.h0) to_bop(df, DF[t], lower);
        }
        float dval = p.dur_t ? (float)__builtin_bit_cast(int, tv_d[t]) : rintf(pr[2][t]);   // torch.round: half to even
        if (p.mask) {                                                                        // networks.py:381-382
            if (rz[t]) dval = 0.0f;
            dval = fmaxf(dval, 0.0f);
        }
        if (p.cum && g == 0) sdur[pos[t]] = rout[t] ? 0 : max((int)dval, 0);
        const unsigned srow = (!rout[t] && g == 0) ? (unsigned)(pos[t] * 4) : kBufOOB;   // one lane per row
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (p.preds[q]) {
                const BufRsrc r_pred = make_rsrc(p.preds[q] + (long)b * p.T, (long)p.T * 4);
                buf_st(r_pred, srow, pr[q][t]);
            }
        }
        if (p.pitch_idx) {
            const BufRsrc r_pi = make_rsrc(p.pitch_idx + (long)b * p.T, (long)p.T * 4);
            buf_st_i(r_pi, srow, bidx[0]);
        }
        if (p.energy_idx) {
            const BufRsrc r_ei = make_rsrc(p.energy_idx + (long)b * p.T, (long)p.T * 4);
            buf_st_i(r_ei, srow, bidx[1]);
        }
        const BufRsrc r_dur = make_rsrc(p.dur + (long)b * p.T, (long)p.T * 4);
        buf_st_i(r_dur, srow, (int)dval);
    }
    // ================================================================ decoder head at phoneme rate: h0 = LN(tanh(feat Wp^T + b)), K = fused | pitch emb |
    // energy emb | duration features (32 channels per step), 256 outputs = 16 tiles of 16; one tile of the wave at a time (its 16 x 4
    // accumulator registers are what the wave can hold), so the 256 KB of weights stream through the two buffers once per tile
    if (p.h0) {
        const BufRsrc r_h0 = make_rsrc(p.h0 + (long)b * p.T * 4 * DIM, (long)p.T * 4 * DIM * 4);
        const BufRsrc r_ep = make_rsrc(p.pred[0].emb, (long)DIM * DIM * 4), r_ee = make_rsrc(p.pred[1].emb, (long)DIM * DIM * 4);
        const int lw8 = wlane(lane, 8);
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
            f32x4 hh[16];
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) hh[nt] = z4;
            f16x2p EM[4];          // the two embedding rows as operands (zero rows for padding phonemes)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned off = rz[t] ? kBufOOB : (unsigned)(hidx[t][q] * DIM * 4) + gl_lane(lane);
                EM[2 * q] = global_bop(q == 0 ? r_ep : r_ee, off, 0);
                EM[2 * q + 1] = global_bop(q == 0 ? r_ep : r_ee, off, 1);
            }
#pragma unroll
            for (int G = 0; G < 8; ++G) {
                const int k = 9 + 8 * t + G;
                step_begin(k);
                const f16x2p op = G < 2 ? F[t][G] : (G < 6 ? EM[G - 2] : DF[t][G - 6]);
                const float* W = wb[k & 1];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    WFrags<4> wf;
                    wfrags_load<4, 8, 4>(wf, 0, W + 2 * c * 256, lw8, 0);
                    f32x4 (&acc4)[4] = *reinterpret_cast<f32x4 (*)[4]>(&hh[4 * c]);
                    mma_all<4>(acc4, wf, op);
                }
            }
            // bias, tanh, LayerNorm over the 256 channels of the row (two-pass), store
            float s1 = 0.0f;
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) {
                const f32x4 hb = ld4_lds(par + VP_HEADB + 16 * nt + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hh[nt][e] = tanh_fast_f32(fmaf(hh[nt][e], kF16WScaleInv, hb[e]));
                    s1 += hh[nt][e];
                }
            }
            const float mean = row_sum4(s1) * (1.0f / 256.0f);
            float s2 = 0.0f;
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hh[nt][e] -= mean;
                    s2 = fmaf(hh[nt][e], hh[nt][e], s2);
                }
            }
            const float rstd = rsqrt_fast_f32(row_sum4(s2) * (1.0f / 256.0f) + 1e-5f);
            const unsigned hrow = rout[t] ? kBufOOB : (unsigned)(pos[t] * 4 * DIM * 4);
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) {
                const f32x4 gg = ld4_lds(par + VP_HEADG + 16 * nt + 4 * g), bb = ld4_lds(par + VP_HEADBE + 16 * nt + 4 * g);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf(hh[nt][e] * rstd, gg[e], bb[e]);
                buf_st4(r_h0, hrow == kBufOOB ? kBufOOB : hrow + (unsigned)((16 * nt + 4 * g) * 4), o);
            }
        }
    }
    if (p.cum) {   // FeatureUpsampler's scan (networks.py:233-244) while the durations are still on the CU; T <= 256 here
        wg_sync_lds();
        if (w == 0) {
            const int per = (p.T + 63) / 64, q0 = lane * per;
            int local = 0;
            for (int q = 0; q < per; ++q) local += (q0 + q < p.T) ? sdur[q0 + q] : 0;
            int incl = local;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = shfl_up_i(incl, d);
                if (lane >= d) incl += v;
            }
            const BufRsrc r_cum = make_rsrc(p.cum + (long)b * p.T, (long)p.T * 4);
            int run = incl - local;
            for (int q = 0; q < per; ++q) {
                run += (q0 + q < p.T) ? sdur[q0 + q] : 0;
                buf_st_i(r_cum, (q0 + q < p.T) ? (unsigned)((q0 + q) * 4) : kBufOOB, run);
            }
            const int total = shfl_i(incl, 63);
            if (lane == 0) p.mel_len[b] = total;
        }
    }
}

template <int NTILE>
__global__ __launch_bounds__(64 * kVa64MaxWaves, 1) void enc_va64_kernel(const FuseVaP p) {
    enc_va64_body<NTILE>(p);
}

}  // namespace esmi
