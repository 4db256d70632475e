// Fuse (layers/networks.py:189-219) of a dim = 128 model with two encoder levels (base ES) over a whole utterance (T <= 256) in one
// workgroup:   fused = mask(Linear_f([ Linear_0(f0) | crop(ConvTranspose1d_{k, stride 2}(Linear_1(f1))) ]))   -> feat[:, 0 .. dim)
// The per-op plan ran it as four GEMM launches with y1, the up-sampled rows and the concatenation through HBM (262 us at B = 512,
// T = 256).  enc_va64.h's plan -- activations in registers, LDS for the weights -- with one change that the wider rows force and that
// also removes its redundant work: a wave owns 16 LEVEL-1 rows n = 16 w + i and the 32 positions they up-sample to as TWO tiles, the
// EVEN positions 2 n and the ODD positions 2 n + 1.  A stride-2 transposed convolution then needs no per-row tap selection:
//     out[2 n]     = W_0 y1[n] + W_2 y1[n - 1] + W_4 y1[n - 2]          out[2 n + 1] = W_1 y1[n] + W_3 y1[n - 1]
// i.e. even taps feed the even tile, odd taps the odd tile, every product is useful, Linear_1 runs once per level-1 row (enc_va64.h:
// twice, and a zero operand for the taps that do not apply), and rows n - 1, n - 2 are DPP row shifts of the y1 tile (+ the last two
// rows of the tile above from a 8 KB LDS exchange buffer).  Everything else is row-wise, so the (even, odd) arrangement is invisible
// outside: the stores go to rows 2 n and 2 n + 1 of feat.
//   * eight waves; products transposed (lane (i, g): row i, channels 16 nt + 4 g + (0..3) of tile nt), GEMM results become the next
//     GEMM's operand in registers (p128::to_bop8);
//   * weights: 5 + k sets of 64 KB (Linear_1: two halves of K = 256; Linear_0; Linear_f's C0 half; the k taps; Linear_f's C1 half -- the
//     order that keeps the fewest rows alive: 192 registers at the peak), two LDS buffers filled by LDS-DMA a step ahead, one workgroup
//     barrier per step.
#pragma once
#include "enc_pred128.h"

namespace esmi {

struct Fuse128Lds {   // floats / dwords
    static constexpr int wbuf = 16 * 1024;                              // one weight set: 64 KB
    static constexpr int w0 = 0, w1 = wbuf, par = 2 * wbuf;
    static constexpr int par_sz = 4 * 128;                              // mlp_b[0], mlp_b[1], up_b[1], fuse_b
    static constexpr int bnd = par + par_sz, bnd_sz = 8 * 2 * 128;      // [tile][row 14 | row 15][k group 4][piece 2][16 dwords]
    static constexpr int total = bnd + bnd_sz;
};
static_assert(Fuse128Lds::total * 4 <= 160 * 1024, "enc_fuse128: LDS");
inline int fuse128_lds_bytes() { return Fuse128Lds::total * (int)sizeof(float); }
enum { FP_B0 = 0, FP_B1 = 128, FP_UPB = 256, FP_FB = 384 };

namespace f128 {
using namespace c16;
using namespace va64;
using namespace p128;
// acc[nt] += W[.., 32 G ..] . X[G]^T over the set's four k groups (W: one 64 KB set in LDS), one tile
__device__ __forceinline__ void set_gemm1(f32x4 (&acc)[NT], const float* W, int lw, const f16x2p* X) {
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
// ... two tiles per weight fragment
__device__ __forceinline__ void set_gemm2(f32x4 (&a0)[NT], f32x4 (&a1)[NT], const float* W, int lw, const f16x2p* X0, const f16x2p* X1) {
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
}  // namespace f128

// KT = the transposed convolution's kernel size (3 or 5)
template <int KT>
__global__ __launch_bounds__(64 * 8, 1) void enc_fuse128_kernel(const FuseVaP p) {
    using namespace c16;
    using namespace va64;
    using namespace p128;
    using namespace f128;
    typedef Fuse128Lds M;
    ESMI_DYN_LDS(lds);
    const int nw = (int)(blockDim.x >> 6), w = uniform_i(wave_id());
    const int lane = lane_id(), i = lane & 15, g = lane >> 4;
    const bool lower = lane < 32;
    const int b = (int)blockIdx.x, rot = (int)blockIdx.x;
    float* const wb[2] = {lds + M::w0, lds + M::w1};
    float* const par = lds + M::par;
    unsigned* const bnd = reinterpret_cast<unsigned*>(lds) + M::bnd;
    const int lw = wlane(lane, 4);
    const f32x4 z4 = zero4();
    const int n1 = p.n_i[1];
    constexpr int NSET = 5 + KT;
    // set k -> buffer k & 1, in the order that keeps the fewest rows alive: Linear_1 (k groups 0..3 | 4..7), Linear_0, Linear_f's C0 half,
    // the KT taps, Linear_f's C1 half
    auto request = [&](int k) __attribute__((always_inline)) {
        const float* src = k < 2 ? p.mlp_w[1] + k * (64 * 256)
                                 : (k == 2 ? p.mlp_w[0] : (k == 3 ? p.fuse_w : (k < 4 + KT ? p.up_w[1] + (k - 4) * (64 * 256) : p.fuse_w + 64 * 256)));
        dma_frags(src, wb[k & 1], 64, w, nw, lane, rot);
    };
    auto step_begin = [&](int k) __attribute__((always_inline)) {
        wait_vm0();
        wg_sync_lds();
        if (k >= 1 && k + 1 < NSET) request(k + 1);
    };
    request(0);
    request(1);
    {   // the four bias vectors: 128 floats = half an instruction each
        const int v2 = lane >> 5, c4 = 4 * (lane & 31);
        if (w == 0 % nw) lds_dma16((v2 ? p.mlp_b[1] : p.mlp_b[0]) + c4, par + FP_B0, lane);
        if (w == 1 % nw) lds_dma16((v2 ? p.fuse_b : p.up_b[1]) + c4, par + FP_UPB, lane);
    }
    const BufRsrc r_mask = make_rsrc(p.mask ? p.mask + (long)b * p.T : nullptr, p.T);
    const BufRsrc r_feat = make_rsrc(p.feat + (long)b * p.T * 4 * DIM, (long)p.T * 4 * DIM * 4);
    const BufRsrc r_f0 = make_rsrc(p.feats[0] + (long)b * p.n_i[0] * DIM, (long)p.n_i[0] * DIM * 4);
    const BufRsrc r_f1 = make_rsrc(p.feats[1] + (long)b * n1 * 2 * DIM, (long)n1 * 2 * DIM * 4);
    const int n = 16 * w + i;                        // this lane's level-1 row; its positions: 2 n (tile 0) and 2 n + 1 (tile 1)
    const bool n_ok = n < n1;
    int pos[2];
    bool rout[2], rz[2];
    f16x2p X0[2][KG];                                // the level-0 rows of the two positions
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        pos[t] = 2 * n + t;
        rout[t] = pos[t] >= p.T;
        rz[t] = !rout[t] && buf_ld_u8(r_mask, (unsigned)pos[t]) != 0;
    }
    // ================================================================ steps 0, 1: y1 = Linear_1(f1 row n) (K = 256: two sets)
    f16x2p Ya[KG];
    {
        f16x2p Xa[2 * KG];
        const unsigned oa = n_ok ? (unsigned)(n * 2 * DIM * 4) + gl_lane(lane) : kBufOOB;
#pragma unroll
        for (int ks = 0; ks < 2 * KG; ++ks) Xa[ks] = global_bop(r_f1, oa, ks);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned o0 = rout[t] ? kBufOOB : (unsigned)(pos[t] * DIM * 4) + gl_lane(lane);
#pragma unroll
            for (int G = 0; G < KG; ++G) X0[t][G] = global_bop(r_f0, o0, G);
        }
        f32x4 a[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a[nt] = z4;
        step_begin(0);
        set_gemm1(a, wb[0], lw, Xa);
        step_begin(1);
        set_gemm1(a, wb[1], lw, Xa + KG);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            a[nt] = n_ok ? fmaf4(a[nt], kF16WScaleInv, ld4_lds(par + FP_B1 + 16 * nt + 4 * g)) : z4;      // rows that do not exist contribute nothing
        to_bop8(a, Ya, lower);
        if (i >= 14) {   // rows 14, 15 of the tile -> the tile below's rows n - 2, n - 1
#pragma unroll
            for (int G = 0; G < KG; ++G) {
                *reinterpret_cast<u32x4*>(bnd + bnd_at8(w, i - 14, G, 0, g)) = Ya[G].h1;
                *reinterpret_cast<u32x4*>(bnd + bnd_at8(w, i - 14, G, 1, g)) = Ya[G].h2;
            }
        }
    }
    // ================================================================ step 2: Linear_0 on the level-0 rows of both tiles; step 3: Linear_f's C0 half
    f32x4 F[2][NT];
    {
        f16x2p C0[2][KG];
        f32x4 a[2][NT];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { a[t][nt] = z4; F[t][nt] = z4; }
        }
        step_begin(2);
        set_gemm2(a[0], a[1], wb[0], lw, X0[0], X0[1]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) a[t][nt] = fmaf4(a[t][nt], kF16WScaleInv, ld4_lds(par + FP_B0 + 16 * nt + 4 * g));
            to_bop8(a[t], C0[t], lower);
        }
        step_begin(3);
        set_gemm2(F[0], F[1], wb[1], lw, C0[0], C0[1]);
    }
    // ================================================================ steps 4 .. 3 + KT: the transposed convolution, tap j -> tile j & 1, rows n - (j >> 1);
    // step 4 + KT: Linear_f's C1 half, masked_fill, store
    {
        f32x4 u[2][NT];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) u[t][nt] = z4;
        }
        f16x2p Ys[KG];         // y1 rows n - s, s = j >> 1: shifted one row further every second tap
#pragma unroll
        for (int G = 0; G < KG; ++G) Ys[G] = Ya[G];
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            step_begin(4 + j);
            if (j >= 2 && (j & 1) == 0) {   // taps 2 s, 2 s + 1 read rows n - s: one more row down, the edge lane takes row 16 - s of the
                const int s = j >> 1;       // tile above (zero above the sequence)
#pragma unroll
                for (int G = 0; G < KG; ++G) Ys[G] = rows_dn(Ys[G], bnd_read8(bnd, w - 1, 2 - s, G, g, w > 0));
            }
            set_gemm1(u[j & 1], wb[(4 + j) & 1], lw, Ys);
        }
        f16x2p C1[2][KG];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) u[t][nt] = fmaf4(u[t][nt], kF16WScaleInv, ld4_lds(par + FP_UPB + 16 * nt + 4 * g));
            to_bop8(u[t], C1[t], lower);
        }
        step_begin(4 + KT);
        set_gemm2(F[0], F[1], wb[(4 + KT) & 1], lw, C1[0], C1[1]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned frow = rout[t] ? kBufOOB : (unsigned)(pos[t] * 4 * DIM * 4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 o = rz[t] ? z4 : fmaf4(F[t][nt], kF16WScaleInv, ld4_lds(par + FP_FB + 16 * nt + 4 * g));
                buf_st4(r_feat, frow == kBufOOB ? kBufOOB : frow + (unsigned)((16 * nt + 4 * g) * 4), o);
            }
        }
    }
}

}  // namespace esmi
