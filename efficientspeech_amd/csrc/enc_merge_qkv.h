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

#ifndef ESMI_E1_WPS
#define ESMI_E1_WPS ESMI_CHAIN_WPS
#endif

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
    float* qkv;            // (B, n_out, 3*h*C)
    int tiles_per_b;       // ceil(n_out / 32)
};

// W'[j][o][i] = sum_m W1[o][m] * Wm[j][m][i]     (fp64 accumulation, once per checkpoint)
__global__ void compose_merge_kernel(const float* __restrict__ wm, const float* __restrict__ w1, float* __restrict__ dst, int k,
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

template <int NCI, int NC>   // Cin = 32*NCI, C = 32*NC
__global__ __launch_bounds__(64, ESMI_E1_WPS) void enc_merge_qkv_kernel(const EncMergeP p) {
    constexpr int CIN = 32 * NCI, C = 32 * NC;
    constexpr int LD = C + 4;
    ESMI_DYN_LDS(buf);             // [32][LD]
    ESMI_CT_INIT(NCI == 4 && NC == 1 ? 3 : 4);
    ESMI_CT();
    const int lane = lane_id(), i = lane & 31, h2 = lane >> 5;
    const int b = (int)blockIdx.x / p.tiles_per_b, tile = (int)blockIdx.x - b * p.tiles_per_b;
    const int t0 = tile * 32, t_out = t0 + i;
    const float* a_row = buf + i * LD + 4 * h2;

    // ---- composed k-tap conv (stride s, zero padding) -> x
    f32x16 x[NC];
    zero_tiles<NC>(x);
    WaveGrp<NC> gc;
    wave_prefetch<NC>(gc, p.merge_w, NC, 0, 0, lane);
    const float* taps[5];   // merge kernels are 1, 3 or 5 wide; masked taps point at a readable row
    bool tok[5];
    int tic[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int ti = t_out * p.stride + j - p.pad;
        tok[j] = j < p.k && t_out < p.n_out && ti >= 0 && ti < p.n_in;
        tic[j] = tok[j] ? ti : 0;
    }
    if (p.ids) {            // block 0: the embedding gather is the conv's A operand
        int id[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) id[j] = p.ids[b * p.n_in + tic[j]];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (id[j] < 0 || id[j] >= p.vocab) id[j] = 0;   // the reference raises IndexError; stay in bounds
            taps[j] = p.table + (long)id[j] * CIN + 4 * h2;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 5; ++j) taps[j] = p.x_in + ((long)b * p.n_in + tic[j]) * CIN + 4 * h2;
    }
    ESMI_CT();
    wave_gemm_taps<NC, 5, NCI, true>(x, gc, taps, tok, p.k, p.merge_w, (long)CIN * C, NC, 0, 0, lane);
    const int nq = 3 * p.h * C, ntq = nq >> 5;
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
