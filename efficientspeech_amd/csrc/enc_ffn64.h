// Everything behind the attention of a C = 64, one-head encoder block whose sequence one workgroup covers (N <= 256: block 0 of small
// ES) in one launch: y1 = mask(LN1(ctx Wp^T + b + x));  out = mask(LN2(mlp2(GELU(conv_k3(y1))) + y1))  (layers/blocks.py:22-29,65,
// layers/networks.py:73-83; the MixFFN Linear is folded into its k = 3 convolution at pack time, esmi.h `ffn_cw`).
// The per-op plan ran this as three GEMM launches with the rows going through HBM between them (87 us at B = 256, N = 256: 100 MB of
// traffic for 0.5 GFLOP each).  Built from enc_va64.h's pieces: a wave owns NTILE 16-row tiles, activations stay in registers (to_bop),
// the convolution's row +-1 operands are DPP row shifts plus a boundary-row exchange, and all 80 KB of weights sit in LDS from the
// entry barrier on (projection + mlp2 in one 48 KB buffer, the convolution's three taps in the other): two workgroup barriers in all.
// In place: `out` may be the buffer `x` (a lane reads its own row of x long before it stores that row of out).
#pragma once
#include "enc_va64.h"

namespace esmi {

struct PostAttn64P {
    const float* ctx;        // (B, N, 64) attention context
    const float* x;          // (B, N, 64) the block's input rows (residual)
    float* out;              // (B, N, 64)
    const float *proj_w, *ffn_w, *mlp2_w;   // esmi_pack_bfrag_f32 arrays (ffn_w: three taps)
    const float *proj_b, *ln1_g, *ln1_b, *ffn_b, *ffn_b0, *ffn_b2, *mlp2_b, *ln2_g, *ln2_b;
    const unsigned char* rowmask;            // (B, N) 1 = padding row, or NULL
    int B, N;
};
struct Ffn64Lds {
    static constexpr int w0 = 0, w1 = Va64Lds::wbuf, par = 2 * Va64Lds::wbuf, par_sz = 768;
    static constexpr int bnd = par + par_sz, total = bnd + Va64Lds::bnd_sz;
};
enum { FP_PROJB = 0, FP_LN1G = 64, FP_LN1B = 128, FP_FFNB = 192, FP_FFNB0 = 256, FP_FFNB2 = 320, FP_MLP2B = 384, FP_LN2G = 448, FP_LN2B = 512 };
inline int ffn64_lds_bytes() { return Ffn64Lds::total * (int)sizeof(float); }

template <int NTILE>
__global__ __launch_bounds__(64 * kVa64MaxWaves, 1) void enc_post_attn64_kernel(const PostAttn64P p) {
    using namespace c16;
    using namespace va64;
    typedef Ffn64Lds M;
    constexpr int C = 64;
    ESMI_DYN_LDS(lds);
    const int nw = (int)(blockDim.x >> 6), w = uniform_i(wave_id());
    const int lane = lane_id(), i = lane & 15, g = lane >> 4;
    const bool lower = lane < 32;
    const int b = (int)blockIdx.x, tile0 = NTILE * w, ntiles = NTILE * nw, rot = (int)blockIdx.x;
    float* const wA = lds + M::w0;
    float* const wB = lds + M::w1;
    float* const par = lds + M::par;
    unsigned* const bnd = reinterpret_cast<unsigned*>(lds) + M::bnd;
    const int lw = wlane(lane, 2);
    const f32x4 z4 = zero4();
    // ---------------- entry: every weight and parameter vector on its way, the rows' own inputs requested
    dma_frags(p.proj_w, wA, 16, w, nw, lane, rot);
    dma_frags(p.mlp2_w, wA + 16 * 256, 16, w, nw, lane, rot);
    dma_frags(p.ffn_w, wB, 48, w, nw, lane, rot);
    {
        const int v4 = lane >> 4, c4 = 4 * (lane & 15);
        auto pick4 = [&](const float* a0, const float* a1, const float* a2, const float* a3) __attribute__((always_inline)) {
            return (v4 & 2 ? (v4 & 1 ? a3 : a2) : (v4 & 1 ? a1 : a0)) + c4;
        };
        if (w == 0 % nw) lds_dma16(pick4(p.proj_b, p.ln1_g, p.ln1_b, p.ffn_b), par + FP_PROJB, lane);
        if (w == 1 % nw) lds_dma16(pick4(p.ffn_b0, p.ffn_b2, p.mlp2_b, p.ln2_g), par + FP_FFNB0, lane);
        if (w == 2 % nw) lds_dma16(pick4(p.ln2_b, p.ln2_b, p.ln2_b, p.ln2_b), par + FP_LN2B, lane);
    }
    const BufRsrc r_ctx = make_rsrc(p.ctx + (long)b * p.N * C, (long)p.N * C * 4);
    const BufRsrc r_x = make_rsrc(p.x + (long)b * p.N * C, (long)p.N * C * 4);
    const BufRsrc r_out = make_rsrc(p.out + (long)b * p.N * C, (long)p.N * C * 4);
    const BufRsrc r_mask = make_rsrc(p.rowmask ? p.rowmask + (long)b * p.N : nullptr, p.N);
    int pos[NTILE];
    bool rout[NTILE], rz[NTILE];
    f16x2p Xc[NTILE][2];
    f32x4 xr[NTILE][4];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        pos[t] = 16 * (tile0 + t) + i;
        rout[t] = pos[t] >= p.N;
        rz[t] = !rout[t] && buf_ld_u8(r_mask, (unsigned)pos[t]) != 0;
        const unsigned row = rout[t] ? kBufOOB : (unsigned)(pos[t] * C * 4);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) Xc[t][ks] = global_bop(r_ctx, row == kBufOOB ? kBufOOB : row + gl_lane(lane), ks);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) xr[t][nt] = buf_ld4(r_x, row == kBufOOB ? kBufOOB : row + (unsigned)((16 * nt + 4 * g) * 4));
    }
    wait_vm0();
    wg_sync_lds();
    // ---------------- y1 = mask(LN1(ctx Wp^T + b + x))
    f32x4 y[NTILE][4];
    f16x2p Y[NTILE][2];
    {
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) y[t][nt] = z4;
        }
        gemm_tiles<NTILE, 2>(y, wA, lw, 0, Xc);
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
            f32x4 gg[4], bb[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                gg[nt] = ld4_lds(par + FP_LN1G + 16 * nt + 4 * g);
                bb[nt] = ld4_lds(par + FP_LN1B + 16 * nt + 4 * g);
                y[t][nt] = fmaf4(y[t][nt], kF16WScaleInv, ld4_lds(par + FP_PROJB + 16 * nt + 4 * g)) + xr[t][nt];
            }
            layernorm<4>(y[t], gg, bb);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                if (rz[t] || rout[t]) y[t][nt] = z4;      // (rows outside the sequence: the convolution's zero padding; never stored)
            to_bop(y[t], Y[t], lower);
        }
        bnd_publish<NTILE>(bnd, tile0, i, g, Y);
    }
    wg_sync_lds();
    // ---------------- MixFFN: (Linear folded into) dense conv k3 -> GELU -> mlp2, residual, LN2, mask
    {
        f32x4 m[NTILE][4], z[NTILE][4];
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { m[t][nt] = z4; z[t][nt] = z4; }
        }
        conv3<NTILE>(m, wB, lw, Y, bnd, tile0, ntiles, g);
        f16x2p Mo[NTILE][2];
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 cb = ld4_lds(par + FP_FFNB + 16 * nt + 4 * g), cb0 = ld4_lds(par + FP_FFNB0 + 16 * nt + 4 * g),
                            cb2 = ld4_lds(par + FP_FFNB2 + 16 * nt + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float bias = cb[e] - (pos[t] == 0 ? cb0[e] : 0.0f) - (pos[t] == p.N - 1 ? cb2[e] : 0.0f);
                    m[t][nt][e] = gelu_fast_f32(fmaf(m[t][nt][e], kF16WScaleInv, bias));
                }
            }
            to_bop(m[t], Mo[t], lower);
        }
        gemm_tiles<NTILE, 2>(z, wA + 16 * 256, lw, 0, Mo);
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
            f32x4 gg[4], bb[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                gg[nt] = ld4_lds(par + FP_LN2G + 16 * nt + 4 * g);
                bb[nt] = ld4_lds(par + FP_LN2B + 16 * nt + 4 * g);
                z[t][nt] = fmaf4(z[t][nt], kF16WScaleInv, ld4_lds(par + FP_MLP2B + 16 * nt + 4 * g)) + y[t][nt];
            }
            layernorm<4>(z[t], gg, bb);
            const unsigned off = rout[t] ? kBufOOB : (unsigned)(pos[t] * C * 4);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) buf_st4(r_out, off == kBufOOB ? kBufOOB : off + (unsigned)((16 * nt + 4 * g) * 4), rz[t] ? z4 : z[t][nt]);
        }
    }
}

}  // namespace esmi
