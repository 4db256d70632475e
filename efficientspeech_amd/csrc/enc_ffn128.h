// Everything behind the attention of a C = 128, two-head, expansion-2 encoder block whose sequence one workgroup covers (N <= 256: block 0
// of base ES) in one launch:   y1 = mask(LN1(ctx Wo^T + b + x));  out = mask(LN2(mlp2(GELU(conv_k3(y1))) + y1))
// (layers/blocks.py:22-29,65, layers/networks.py:73-83; ctx = the per-head P_h x of the weight-folded attention, Wo = [Wv_h^T Wp_h^T] (esmi.h
// `vo_w`), the MixFFN Linear folded into its k = 3 convolution (`ffn_cw`)).  The per-op plan ran this as three GEMM launches with y1 and the
// 256-wide hidden rows through HBM between them: 369 us at B = 512, N = 256 for 43 GFLOP (1.4 - 2.7 TB/s, 116 TFLOP/s).
// enc_ffn64.h's plan at twice the width -- activations in registers, LDS for the weights -- needs the register file twice, so:
//   * eight waves x two 16-row tiles.  The projection (K = 256 -> 128, two 64 KB sets) runs on both tiles per weight fragment; y1 goes to a
//     scratch tensor (the residual of LN2, re-read by the lane that wrote it) and stays in registers as the convolution's operand;
//   * the FFN then runs ONE TILE AT A TIME (the 256 hidden channels of two tiles do not fit next to y1): per tile the convolution's three
//     taps for hidden channels 0..127, GELU -> the first half of mlp2's operand, the three taps for channels 128..255, GELU -> the second
//     half, mlp2 (K = 256 -> 128, two sets), + y1, LN2, store: eight 64 KB weight sets per tile, 18 in all through the two LDS buffers
//     (LDS-DMA a step ahead, one workgroup barrier per step).  A set of the convolution = (tap, half of the output channels): its
//     fragments are 16 runs of 4 KB in the packed array (esmi_pack_bfrag_f32: [tap][k group][slot][8 row tiles]) and land as a 128-row
//     matrix;
//   * row +-1 operands: DPP row shifts + the boundary rows of all 16 tiles in a 16 KB LDS exchange buffer (enc_pred128.h).
#pragma once
#include "enc_pred128.h"

namespace esmi {

struct Ffn128Lds {   // floats / dwords
    static constexpr int wbuf = 16 * 1024;
    static constexpr int w0 = 0, w1 = wbuf, par = 2 * wbuf, par_sz = 6 * 128 + 3 * 256;
    static constexpr int bnd = par + par_sz, bnd_sz = 16 * 2 * 128, total = bnd + bnd_sz;
};
static_assert(Ffn128Lds::total * 4 <= 160 * 1024, "enc_ffn128: LDS");
inline int ffn128_lds_bytes() { return Ffn128Lds::total * (int)sizeof(float); }
enum { GP_PROJB = 0, GP_LN1G = 128, GP_LN1B = 256, GP_MLP2B = 384, GP_LN2G = 512, GP_LN2B = 640, GP_FFNB = 768, GP_FFNB0 = 1024, GP_FFNB2 = 1280 };

namespace g128 {
using namespace c16;
using namespace va64;
using namespace p128;
// one tap of a k = 3 convolution on ONE tile: c[nt] += W_j . X^T(row + j - 1), W = (tap, output half) as a 128-row matrix in LDS
template <int J>
__device__ __forceinline__ void conv_tap1(f32x4 (&c)[NT], const float* W, int lw, const f16x2p (&X)[KG], const unsigned* bnd, int tile, int ntiles, int g) {
#pragma unroll
    for (int G = 0; G < KG; ++G) {
        f16x2p op;
        if (J == 0) op = rows_dn(X[G], bnd_read8(bnd, tile - 1, 1, G, g, tile > 0));
        else if (J == 1) op = X[G];
        else op = rows_up(X[G], bnd_read8(bnd, tile + 1, 0, G, g, tile + 1 < ntiles));
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            WFrags<4> wf;
            wfrags_load<4, 4, 4>(wf, 0, W + 2 * ch * 256, lw, G);
            f32x4 (&acc4)[4] = *reinterpret_cast<f32x4 (*)[4]>(&c[4 * ch]);
            mma_all<4>(acc4, wf, op);
        }
    }
}
// acc[nt] += W[.., 32 G ..] . X[G]^T over the set's four k groups, one tile / two tiles per weight fragment
__device__ __forceinline__ void set1(f32x4 (&acc)[NT], const float* W, int lw, const f16x2p* X) {
#pragma unroll
    for (int G = 0; G < KG; ++G) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            WFrags<4> wf;
            wfrags_load<4, 4, 4>(wf, 0, W + 2 * ch * 256, lw, G);
            f32x4 (&acc4)[4] = *reinterpret_cast<f32x4 (*)[4]>(&acc[4 * ch]);
            mma_all<4>(acc4, wf, X[G]);
        }
    }
}
__device__ __forceinline__ void set2(f32x4 (&a0)[NT], f32x4 (&a1)[NT], const float* W, int lw, const f16x2p* X0, const f16x2p* X1) {
#pragma unroll
    for (int G = 0; G < KG; ++G) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            WFrags<4> wf;
            wfrags_load<4, 4, 4>(wf, 0, W + 2 * ch * 256, lw, G);
            f32x4 (&p0)[4] = *reinterpret_cast<f32x4 (*)[4]>(&a0[4 * ch]);
            f32x4 (&p1)[4] = *reinterpret_cast<f32x4 (*)[4]>(&a1[4 * ch]);
            mma_all<4>(p0, wf, X0[G]);
            mma_all<4>(p1, wf, X1[G]);
        }
    }
}
}  // namespace g128

__global__ __launch_bounds__(64 * 8, 1) void enc_post_attn128_kernel(const PostAttn128P p) {
    using namespace c16;
    using namespace va64;
    using namespace p128;
    using namespace g128;
    typedef Ffn128Lds M;
    constexpr int C = 128;
    ESMI_DYN_LDS(lds);
    const int nw = (int)(blockDim.x >> 6), w = uniform_i(wave_id());
    const int lane = lane_id(), i = lane & 15, g = lane >> 4;
    const bool lower = lane < 32;
    const int b = (int)blockIdx.x, tile0 = 2 * w, ntiles = 2 * nw, rot = (int)blockIdx.x;
    float* const wb[2] = {lds + M::w0, lds + M::w1};
    float* const par = lds + M::par;
    unsigned* const bnd = reinterpret_cast<unsigned*>(lds) + M::bnd;
    const int lw = wlane(lane, 4);
    const f32x4 z4 = zero4();
    constexpr int NSET = 18;
    // set k -> buffer k & 1: 0, 1 = the projection's k groups 0..3 | 4..7; then per tile eight sets: the convolution's taps 0, 1, 2 for hidden
    // channels 0..127, the same for 128..255, mlp2's k groups 0..3 | 4..7
    auto request = [&](int k) __attribute__((always_inline)) {
        float* dst = wb[k & 1];
        if (k < 2) {
            dma_frags(p.proj_w + k * (64 * 256), dst, 64, w, nw, lane, rot);
        } else {
            const int r = (k - 2) & 7;
            if (r >= 6) {
                dma_frags(p.mlp2_w + (r - 6) * (64 * 256), dst, 64, w, nw, lane, rot);
            } else {
                const int tap = r % 3, half = r / 3;
                const float* src = p.ffn_w + tap * (128 * 256) + half * (4 * 256);
                for (int f = w; f < 64; f += nw) {           // fragment (k group, slot, row tile) of the 128-row matrix <- row tile 4 half + tile of the 256-row one
                    const int fr = (f + rot) & 63;
                    lds_dma16(src + ((fr >> 2) * 8 + (fr & 3)) * 256 + 4 * lane, dst + fr * 256, lane);
                }
            }
        }
    };
    auto step_begin = [&](int k) __attribute__((always_inline)) {
        wait_vm0();
        wg_sync_lds();
        if (k >= 1 && k + 1 < NSET) request(k + 1);
    };
    request(0);
    request(1);
    {   // parameter vectors: six of 128 floats (half an instruction each), three of 256 (one instruction each)
        const int v2 = lane >> 5, c4 = 4 * (lane & 31);
        if (w == 0 % nw) lds_dma16((v2 ? p.ln1_g : p.proj_b) + c4, par + GP_PROJB, lane);
        if (w == 1 % nw) lds_dma16((v2 ? p.mlp2_b : p.ln1_b) + c4, par + GP_LN1B, lane);
        if (w == 2 % nw) lds_dma16((v2 ? p.ln2_b : p.ln2_g) + c4, par + GP_LN2G, lane);
        if (w == 3 % nw) lds_dma16(p.ffn_b + 4 * lane, par + GP_FFNB, lane);
        if (w == 4 % nw) lds_dma16(p.ffn_b0 + 4 * lane, par + GP_FFNB0, lane);
        if (w == 5 % nw) lds_dma16(p.ffn_b2 + 4 * lane, par + GP_FFNB2, lane);
    }
    const BufRsrc r_ctx = make_rsrc(p.ctx + (long)b * p.N * 2 * C, (long)p.N * 2 * C * 4);
    const BufRsrc r_x = make_rsrc(p.x + (long)b * p.N * C, (long)p.N * C * 4);
    const BufRsrc r_y1 = make_rsrc(p.y1 + (long)b * p.N * C, (long)p.N * C * 4);
    const BufRsrc r_out = make_rsrc(p.out + (long)b * p.N * C, (long)p.N * C * 4);
    const BufRsrc r_mask = make_rsrc(p.rowmask ? p.rowmask + (long)b * p.N : nullptr, p.N);
    int pos[2];
    bool rout[2], rz[2];
    unsigned row_c[2], row_x[2];     // byte offsets of the lane's rows (ctx: + this lane's operand chunk), kBufOOB outside the sequence
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        pos[t] = 16 * (tile0 + t) + i;
        rout[t] = pos[t] >= p.N;
        rz[t] = !rout[t] && buf_ld_u8(r_mask, (unsigned)pos[t]) != 0;
        row_c[t] = rout[t] ? kBufOOB : (unsigned)(pos[t] * 2 * C * 4) + gl_lane(lane);
        row_x[t] = rout[t] ? kBufOOB : (unsigned)(pos[t] * C * 4);
    }
    // ================================================================ steps 0, 1: y1 = mask(LN1(ctx Wo^T + b + x)) on both tiles
    f16x2p Y[2][KG];
    {
        f32x4 a[2][NT];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) a[t][nt] = z4;
        }
        f16x2p Xc[2][KG];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int G = 0; G < KG; ++G) Xc[t][G] = global_bop(r_ctx, row_c[t], G);
        }
        step_begin(0);
        set2(a[0], a[1], wb[0], lw, Xc[0], Xc[1]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int G = 0; G < KG; ++G) Xc[t][G] = global_bop(r_ctx, row_c[t], KG + G);
        }
        step_begin(1);
        set2(a[0], a[1], wb[1], lw, Xc[0], Xc[1]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 gg[NT], bb[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 xr = buf_ld4(r_x, row_x[t] == kBufOOB ? kBufOOB : row_x[t] + (unsigned)((16 * nt + 4 * g) * 4));
                gg[nt] = ld4_lds(par + GP_LN1G + 16 * nt + 4 * g);
                bb[nt] = ld4_lds(par + GP_LN1B + 16 * nt + 4 * g);
                a[t][nt] = fmaf4(a[t][nt], kF16WScaleInv, ld4_lds(par + GP_PROJB + 16 * nt + 4 * g)) + xr;
            }
            layernorm<NT>(a[t], gg, bb);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (rz[t] || rout[t]) a[t][nt] = z4;      // (rows outside the sequence: the convolution's zero padding; never stored)
                buf_st4(r_y1, row_x[t] == kBufOOB ? kBufOOB : row_x[t] + (unsigned)((16 * nt + 4 * g) * 4), a[t][nt]);
            }
            to_bop8(a[t], Y[t], lower);
        }
        bnd_publish8(bnd, tile0, i, g, Y);
    }
    // ================================================================ per tile: conv (two halves of the hidden channels) -> GELU -> mlp2, + y1, LN2, mask
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int k0 = 2 + 8 * t;
        f16x2p Mo[2][KG];          // mlp2's operand: hidden channels 0..127 | 128..255
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            f32x4 m[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) m[nt] = z4;
            step_begin(k0 + 3 * hf);
            conv_tap1<0>(m, wb[(k0 + 3 * hf) & 1], lw, Y[t], bnd, tile0 + t, ntiles, g);
            step_begin(k0 + 3 * hf + 1);
            conv_tap1<1>(m, wb[(k0 + 3 * hf + 1) & 1], lw, Y[t], bnd, tile0 + t, ntiles, g);
            step_begin(k0 + 3 * hf + 2);
            conv_tap1<2>(m, wb[(k0 + 3 * hf + 2) & 1], lw, Y[t], bnd, tile0 + t, ntiles, g);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c0 = 128 * hf + 16 * nt + 4 * g;
                const f32x4 cb = ld4_lds(par + GP_FFNB + c0), cb0 = ld4_lds(par + GP_FFNB0 + c0), cb2 = ld4_lds(par + GP_FFNB2 + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float bias = cb[e] - (pos[t] == 0 ? cb0[e] : 0.0f) - (pos[t] == p.N - 1 ? cb2[e] : 0.0f);
                    m[nt][e] = gelu_fast_f32(fmaf(m[nt][e], kF16WScaleInv, bias));
                }
            }
            to_bop8(m, Mo[hf], lower);
        }
        f32x4 z[NT], yr[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) z[nt] = z4;
        step_begin(k0 + 6);
        set1(z, wb[(k0 + 6) & 1], lw, Mo[0]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) yr[nt] = buf_ld4(r_y1, row_x[t] == kBufOOB ? kBufOOB : row_x[t] + (unsigned)((16 * nt + 4 * g) * 4));   // (written by this lane in step 1)
        step_begin(k0 + 7);
        set1(z, wb[(k0 + 7) & 1], lw, Mo[1]);
        f32x4 gg[NT], bb[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            gg[nt] = ld4_lds(par + GP_LN2G + 16 * nt + 4 * g);
            bb[nt] = ld4_lds(par + GP_LN2B + 16 * nt + 4 * g);
            z[nt] = fmaf4(z[nt], kF16WScaleInv, ld4_lds(par + GP_MLP2B + 16 * nt + 4 * g)) + yr[nt];
        }
        layernorm<NT>(z, gg, bb);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            buf_st4(r_out, row_x[t] == kBufOOB ? kBufOOB : row_x[t] + (unsigned)((16 * nt + 4 * g) * 4), rz[t] ? z4 : z[nt]);
    }
}

}  // namespace esmi
