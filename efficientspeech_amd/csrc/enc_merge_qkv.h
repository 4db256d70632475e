// Fused first half of an encoder block (layers/networks.py:54,64-67 + layers/blocks.py:44):
//     x   = Conv1x1( Conv1d_k,stride( Embedding(ids) | x_in ) )        both convs dense and bias-free
//     qkv = Linear(C, 3*h*C, bias=False)(x)
// The two merge convolutions are linear with nothing in between, so they are applied as ONE k-tap convolution
// Cin -> C whose weights W'[j] = W_1x1 . W_merge[j] are composed once per checkpoint (esmi_compose_merge_f32, fp64
// accumulation): tiny ES block 0 needs 192 instead of 832 MFMAs per tile.  One wave per (utterance, 32-position
// tile); the conv reads its shifted input rows straight from global memory / the embedding table, the qkv GEMM
// takes its A fragments from the wave's LDS tile.
#pragma once
#include "wave_chain.h"

namespace esmi {

struct EncMergeP {
    const int* ids;        // block 0: (B, n_in) phoneme ids, else NULL
    const float* table;    // block 0: (vocab, Cin) embedding
    int vocab;
    const float* x_in;     // blocks >= 1: (B, n_in, Cin)
    int B, n_in, n_out, k, stride, pad, h;
    const float* merge_w;  // (k, C, Cin) composed merge conv  } both in MFMA B-fragment order
    const float* qkv_w;    // (3*h*C, C)                        } (esmi_pack_bfrag_f32, see wave_chain.h)
    float* x_out;          // (B, n_out, C)
    float* qkv;            // (B, n_out, 3*h*C)  [nq_override > 0: (B, n_out, nq_override), with qkv_w (nq_override, C)]
    int nq_override;       // 0, or the width of the Linear behind the merge convs when it is not the reference's qkv (folded attention: h*C)
    int tiles_per_b;       // ceil(n_out / 32)
    const float* emb_conv; // block 0, optional: (k, vocab, C) = embedding folded into the composed merge conv (esmi.h), else NULL
};

// W'[j][o][i] = sum_m W1[o][m] * Wm[j][m][i]     (fp64 accumulation, once per checkpoint)
static __global__ void compose_merge_kernel(const float* __restrict__ wm, const float* __restrict__ w1, float* __restrict__ dst, int k,
                                     int cin, int cout) {
    const long n = (long)k * cout * cin;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int i = (int)(e % cin);
        const int o = (int)((e / cin) % cout);
        const int j = (int)(e / ((long)cin * cout));
        double acc = 0.0;
        for (int m = 0; m < cin; ++m) acc += (double)w1[(long)o * cin + m] * (double)wm[((long)j * cin + m) * cin + i];
        dst[e] = (float)acc;
    }
}

inline int enc_merge_lds_floats(int c_in, int c_out, int k, int stride) {
    const int in_t = (31 * stride + k) * (c_in + 4), x_t = 32 * (c_out + 4);
    return in_t > x_t ? in_t : x_t;
}

// Stage the input rows under one 32-row output tile in `stg` ([31*STRIDE + KT][Cin + 4], private to the wave) with
// full-row coalesced loads, then run the composed k-tap conv: x = tile rows [t0, t0 + 32) of utterance b.
// (Reading the A fragments straight from global memory -- 32 rows x 32 bytes per instruction, every 128-byte line
// touched by four instructions -- cost more than the MFMAs of the composed conv.)  Rows outside [0, n_in) are the
// conv's zero padding.
template <int NCI, int NC, int KT, int STRIDE>
__device__ __forceinline__ void merge_conv_tile(const EncMergeP& p, int b, int t0, float* stg, int lane, f32x16 (&x)[NC]) {
    constexpr int CIN = 32 * NCI, C = 32 * NC;
    constexpr int LDI = CIN + 4;
    constexpr int LPR = 8 * NCI;               // lanes per input row (16 bytes each)
    constexpr int RPI = 64 / LPR;              // input rows per wave-wide load
    constexpr int NROWS = 31 * STRIDE + KT;    // input rows under one 32-row output tile
    constexpr int MAXI = (NROWS + RPI - 1) / RPI;
    const int i = lane & 31, h2 = lane >> 5;
    if (p.ids && p.emb_conv) {
        // block 0 with the embedding folded in (esmi.h, emb_conv): x[t] = sum_j E_j[id[t*STRIDE + j - pad]] -- two dependent round trips
        // (ids, then KT rows of C floats per position) straight into the accumulator layout; no staging tile, no contraction
        const int* idb = p.ids + (long)b * p.n_in;
        int id[16][KT];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                const int ti = (t0 + tile_row(r, lane)) * STRIDE + j - p.pad;
                const bool ok = ti >= 0 && ti < p.n_in;
                const int v = idb[ok ? ti : 0];
                id[r][j] = !ok ? -1 : ((v < 0 || v >= p.vocab) ? 0 : v);   // (the reference raises IndexError; stay in bounds)
            }
        }
        zero_tiles<NC>(x);
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const float* ej = p.emb_conv + (long)j * p.vocab * C + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* row = ej + (long)(id[r][j] < 0 ? 0 : id[r][j]) * C;
#pragma unroll
                for (int nt = 0; nt < NC; ++nt) {
                    const float v = row[32 * nt];
                    x[nt][r] += id[r][j] < 0 ? 0.0f : v;
                }
            }
        }
        return;
    }
    WaveGrp<NC> gc;
    wave_prefetch<NC>(gc, p.merge_w, NC, 0, 0, lane);
    const int ti0 = t0 * STRIDE - p.pad;        // input position of staged row 0
    const int lrow = lane / LPR, lchunk = lane - lrow * LPR;
    f32x4 sv[MAXI];
    if (p.ids) {                   // block 0: the embedding gather is the conv's input
        int id[MAXI];
#pragma unroll
        for (int m = 0; m < MAXI; ++m) {
            const int ti = ti0 + m * RPI + lrow;
            id[m] = p.ids[b * p.n_in + (ti >= 0 && ti < p.n_in ? ti : 0)];
        }
#pragma unroll
        for (int m = 0; m < MAXI; ++m) {
            const int ti = ti0 + m * RPI + lrow;
            if (id[m] < 0 || id[m] >= p.vocab) id[m] = 0;   // the reference raises IndexError; stay in bounds
            sv[m] = ld4(p.table + (long)id[m] * CIN + 4 * lchunk);
            if (ti < 0 || ti >= p.n_in) sv[m] = zero4();
        }
    } else {
        const BufRsrc r_in = make_rsrc(p.x_in + (long)b * p.n_in * CIN, (long)p.n_in * CIN * 4);
#pragma unroll
        for (int m = 0; m < MAXI; ++m) {
            const int ti = ti0 + m * RPI + lrow;
            sv[m] = buf_ld4(r_in, (unsigned)((ti * CIN + 4 * lchunk) * 4));   // out of range reads 0
        }
    }
#pragma unroll
    for (int m = 0; m < MAXI; ++m) {
        const int rr = m * RPI + lrow;
        if (rr < NROWS) *reinterpret_cast<f32x4*>(stg + rr * LDI + 4 * lchunk) = sv[m];
    }
    lds_wave_sync();
    zero_tiles<NC>(x);
    const float* taps[KT];
    bool tok[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        taps[j] = stg + (i * STRIDE + j) * LDI + 4 * h2;
        tok[j] = true;
    }
    wave_gemm_taps<NC, KT, NCI, false>(x, gc, taps, tok, p.merge_w, (long)CIN * C, NC, 0, 0, lane);
}

template <int NCI, int NC, int KT, int STRIDE>   // Cin = 32*NCI, C = 32*NC, merge kernel KT, stride STRIDE
__global__ __launch_bounds__(64, kChainWps) void enc_merge_qkv_kernel(const EncMergeP p) {
    constexpr int C = 32 * NC;
    constexpr int LD = C + 4;
    ESMI_DYN_LDS(buf);             // input rows [31*STRIDE + KT][Cin + 4], later the x tile [32][LD]
    ESMI_CT_INIT(NCI == 4 && NC == 1 ? 3 : 4);
    ESMI_CT();
    const int lane = lane_id(), i = lane & 31, h2 = lane >> 5;
    const int b = (int)blockIdx.x / p.tiles_per_b, tile = (int)blockIdx.x - b * p.tiles_per_b;
    const int t0 = tile * 32;
    const float* a_row = buf + i * LD + 4 * h2;
    f32x16 x[NC];
    merge_conv_tile<NCI, NC, KT, STRIDE>(p, b, t0, buf, lane, x);
    const int nq = p.nq_override > 0 ? p.nq_override : 3 * p.h * C, ntq = nq >> 5;
    WaveGrp<4> gq;
    wave_prefetch<4>(gq, p.qkv_w, ntq, 0, 0, lane);
    ESMI_CT();
    // rows >= n_out fall off the end of the per-utterance buffers: the stores need no branch
    const BufRsrc r_xo = make_rsrc(p.x_out + (long)b * p.n_out * C, (long)p.n_out * C * 4);
    const BufRsrc r_qkv = make_rsrc(p.qkv + (long)b * p.n_out * nq, (long)p.n_out * nq * 4);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned off = (unsigned)(((t0 + tile_row(r, lane)) * C + i) * 4);
#pragma unroll
        for (int nt = 0; nt < NC; ++nt) buf_st(r_xo, off + 128u * nt, x[nt][r]);
    }
    lds_wave_sync();               // the conv has read its last input row
    tile_store<NC>(buf, LD, 0, x, lane);
    lds_wave_sync();
    ESMI_CT();
    // ---- qkv, 128 output channels per pass
    for (int n0 = 0; n0 < nq; n0 += 128) {
        f32x16 q[4];
        zero_tiles<4>(q);
        wave_gemm_k<4, NC>(q, gq, a_row, true, p.qkv_w, ntq, 0, n0 >> 5, lane);
        if (n0 + 128 < nq) wave_prefetch<4>(gq, p.qkv_w, ntq, 0, (n0 + 128) >> 5, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned off = (unsigned)(((t0 + tile_row(r, lane)) * nq + n0 + i) * 4);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) buf_st(r_qkv, n0 + 32 * nt < nq ? off + 128u * nt : kBufOOB, q[nt][r]);
        }
        ESMI_CT();
    }
}

}  // namespace esmi
