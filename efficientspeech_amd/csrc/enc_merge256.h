// The front of a C = 256 encoder block fed by 128-channel rows (block 1 of base ES: k = 3, four heads, N <= 128 output rows) in one launch:
//     x1 = Conv1d_{k, stride 2, pad k/2}(x0; the block's two merge convolutions composed at pack time, 128 -> 256, bias-free)        (networks.py:64-67)
//     q  = x1 [M_0 | .. | M_{H-1}],  M_h = Wq_h^T Wk_h  (the weight-folded attention's query side, 256 -> H x 256; keys = values = x1)   (blocks.py:44)
// The per-op plan ran the strided convolution on the un-staged GEMM kernel (105 us) and the query GEMM as its own launch with x1 through HBM
// (233 us): 338 us at B = 512 for 47 GFLOP.  Here a wave owns 16 OUTPUT rows n and the input rows 2 n (tile E) and 2 n + 1 (tile O):
//     k = 3:  x1[n] = W_0 O[n-1] + W_1 E[n] + W_2 O[n]          k = 5:  x1[n] = W_0 E[n-1] + W_1 O[n-1] + W_2 E[n] + W_3 O[n] + W_4 E[n+1]
// -- every tap is a full product on an unstrided tile, rows n -+ 1 are DPP row shifts + an LDS exchange of the tiles' first / last rows
// (enc_fuse128.h's even / odd trick read backwards) -- x1 becomes the query GEMM's operand in registers, and all weights go through the
// two 64 KB LDS buffers as 2 k + 4 H sets (LDS-DMA a step ahead, one barrier per step): (tap, half of x1's channels), then (128 of q's 256 H
// channels, half of K).  Activations never touch LDS.
#pragma once
#include "enc_pred128.h"

namespace esmi {

struct Merge256Lds {   // floats / dwords
    static constexpr int wbuf = 16 * 1024;
    static constexpr int w0 = 0, w1 = wbuf, bnd = 2 * wbuf, bnd_sz = 16 * 2 * 128, total = bnd + bnd_sz;
};
static_assert(Merge256Lds::total * 4 <= 160 * 1024, "enc_merge256: LDS");
inline int merge256_lds_bytes() { return Merge256Lds::total * (int)sizeof(float); }

template <int KT, int H>      // merge kernel size (3 or 5), heads
__global__ __launch_bounds__(64 * 8, 1) void enc_merge_q256_kernel(const MergeQ256P p) {
    using namespace c16;
    using namespace va64;
    using namespace p128;
    typedef Merge256Lds M;
    constexpr int CI = 128, CO = 256;
    ESMI_DYN_LDS(lds);
    const int nw = (int)(blockDim.x >> 6), w = uniform_i(wave_id());
    const int lane = lane_id(), i = lane & 15, g = lane >> 4;
    const bool lower = lane < 32;
    const int b = (int)blockIdx.x, rot = (int)blockIdx.x;
    float* const wb[2] = {lds + M::w0, lds + M::w1};
    unsigned* const bnd = reinterpret_cast<unsigned*>(lds) + M::bnd;
    const int lw = wlane(lane, 4);
    const f32x4 z4 = zero4();
    constexpr int NM = 2 * KT, NSET = NM + 4 * H, PADK = KT / 2;
    // a set = 64 fragments (k group, slot, row tile) of a 128-row matrix, cut out of a packed array with `ntw` row tiles per (k group, slot):
    // fragment fr <- src[((fr >> 2) * ntw + (fr & 3)) * 256]
    auto dma_cut = [&](const float* src, int ntw, float* dst) __attribute__((always_inline)) {
        for (int f = w; f < 64; f += nw) {
            const int fr = (f + rot) & 63;
            lds_dma16(src + ((fr >> 2) * ntw + (fr & 3)) * 256 + 4 * lane, dst + fr * 256, lane);
        }
    };
    // set k -> buffer k & 1: k < NM: tap k % KT of the merge convolution, x1 channels 128 (k / KT) ..; then (q channels 128 pq .., k groups 4 kh ..): k = NM + 2 pq + kh
    auto request = [&](int k) __attribute__((always_inline)) {
        if (k < NM) dma_cut(p.merge_w + (k % KT) * (CI / 8 * 8 * 256) + (k / KT) * (4 * 256), 8, wb[k & 1]);
        else dma_cut(p.q_w + ((k - NM) & 1) * (16 * 8 * H * 256) + ((k - NM) >> 1) * (4 * 256), 8 * H, wb[k & 1]);
    };
    auto step_begin = [&](int k) __attribute__((always_inline)) {
        wait_vm0();
        wg_sync_lds();
        if (k >= 1 && k + 1 < NSET) request(k + 1);
    };
    request(0);
    request(1);
    const BufRsrc r_in = make_rsrc(p.x_in + (long)b * p.n_in * CI, (long)p.n_in * CI * 4);
    const BufRsrc r_x1 = make_rsrc(p.x_out + (long)b * p.n_out * CO, (long)p.n_out * CO * 4);
    const BufRsrc r_q = make_rsrc(p.q + (long)b * p.n_out * H * CO, (long)p.n_out * H * CO * 4);
    const int n = 16 * w + i;                         // this lane's output row; its input rows 2 n (E) and 2 n + 1 (O); rows >= n_in read as zeros
    const bool rout = n >= p.n_out;
    f16x2p E[KG], O[KG];
    {
        const unsigned oe = (unsigned)(2 * n * CI * 4) + gl_lane(lane), oo = oe + (unsigned)(CI * 4);
#pragma unroll
        for (int G = 0; G < KG; ++G) {
            E[G] = global_bop(r_in, 2 * n < p.n_in ? oe : kBufOOB, G);
            O[G] = global_bop(r_in, 2 * n + 1 < p.n_in ? oo : kBufOOB, G);
        }
    }
    // boundary rows: E's first / last row of tile w at (w, side 0 / 1), O's last row at (8 + w, side 1)
    if (i == 0 || i == 15) {
#pragma unroll
        for (int G = 0; G < KG; ++G) {
            *reinterpret_cast<u32x4*>(bnd + bnd_at8(w, i == 0 ? 0 : 1, G, 0, g)) = E[G].h1;
            *reinterpret_cast<u32x4*>(bnd + bnd_at8(w, i == 0 ? 0 : 1, G, 1, g)) = E[G].h2;
            if (i == 15) {
                *reinterpret_cast<u32x4*>(bnd + bnd_at8(8 + w, 1, G, 0, g)) = O[G].h1;
                *reinterpret_cast<u32x4*>(bnd + bnd_at8(8 + w, 1, G, 1, g)) = O[G].h2;
            }
        }
    }
    // ================================================================ steps 0 .. NM - 1: x1 = the strided convolution, 128 channels at a time
    f16x2p X1[2 * KG];
    const unsigned xrow = rout ? kBufOOB : (unsigned)(n * CO * 4), qrow = rout ? kBufOOB : (unsigned)(n * H * CO * 4);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        f32x4 a[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a[nt] = z4;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const int k = KT * hf + j;
            const int d = j - PADK;                      // input row 2 n + d: d even -> E[n + d / 2], d odd -> O[n + (d - 1) / 2]
            step_begin(k);
            const float* W = wb[k & 1];
#pragma unroll
            for (int G = 0; G < KG; ++G) {
                f16x2p op;
                if (d == -2) op = rows_dn(E[G], bnd_read8(bnd, w - 1, 1, G, g, w > 0));
                else if (d == -1) op = rows_dn(O[G], bnd_read8(bnd, 8 + w - 1, 1, G, g, w > 0));
                else if (d == 0) op = E[G];
                else if (d == 1) op = O[G];
                else op = rows_up(E[G], bnd_read8(bnd, w + 1, 0, G, g, w + 1 < nw));
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    WFrags<4> wf;
                    wfrags_load<4, 4, 4>(wf, 0, W + 2 * ch * 256, lw, G);
                    f32x4 (&acc4)[4] = *reinterpret_cast<f32x4 (*)[4]>(&a[4 * ch]);
                    mma_all<4>(acc4, wf, op);
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[nt][e] *= kF16WScaleInv;
            buf_st4(r_x1, xrow == kBufOOB ? kBufOOB : xrow + (unsigned)((128 * hf + 16 * nt + 4 * g) * 4), a[nt]);
        }
        f16x2p hb[KG];
        to_bop8(a, hb, lower);
#pragma unroll
        for (int G = 0; G < KG; ++G) X1[KG * hf + G] = hb[G];
    }
    // ================================================================ the last 4 H steps: q = x1 [M_0 | ..], 128 output channels at a time, K = 256 in two sets
#pragma unroll
    for (int pq = 0; pq < 2 * H; ++pq) {
        f32x4 a[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a[nt] = z4;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int k = NM + 2 * pq + kh;
            step_begin(k);
            const float* W = wb[k & 1];
#pragma unroll
            for (int G = 0; G < KG; ++G) {
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    WFrags<4> wf;
                    wfrags_load<4, 4, 4>(wf, 0, W + 2 * ch * 256, lw, G);
                    f32x4 (&acc4)[4] = *reinterpret_cast<f32x4 (*)[4]>(&a[4 * ch]);
                    mma_all<4>(acc4, wf, X1[KG * kh + G]);
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[nt][e] *= kF16WScaleInv;
            buf_st4(r_q, qrow == kBufOOB ? kBufOOB : qrow + (unsigned)((128 * pq + 16 * nt + 4 * g) * 4), a[nt]);
        }
    }
}

}  // namespace esmi
