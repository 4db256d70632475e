// wave_chain -- building blocks for the fused encoder-side kernels.
//
// A "chain" kernel gives ONE wave (a 64-thread workgroup) a 32-row tile of an utterance and carries it
// through several dependent GEMMs without leaving the CU: the MFMA C/D accumulators of one stage are
// written to a private LDS tile [rows][K+4] and re-read as the next stage's A fragments (lane
// (i = lane&31, h = lane>>5) reads 16 bytes: channels [8kc+4h, +4) of row i).  Convolutions along the
// sequence are row shifts of the A fragment inside the tile (halo rows are recomputed by neighbouring
// tiles); weights stream from L2 as 16-byte pieces, four k-steps per round trip.
#pragma once
#include "esmi_dev.h"

#ifdef ESMI_CHAIN_TRACE
// development only: shader-clock stamps of workgroup 7 of each chain kernel -> g_chain_trace[kernel_slot*64 + n]
extern __device__ long long* g_chain_trace_dev;
#define ESMI_CT_INIT(slot) int ct_n_ = 0; const bool ct_on_ = blockIdx.x == 7 && (threadIdx.x & 63) == 0; const int ct_slot_ = (slot)
#define ESMI_CT() do { if (ct_on_ && g_chain_trace_dev) g_chain_trace_dev[ct_slot_ * 64 + ct_n_] = (long long)__builtin_amdgcn_s_memtime(); ++ct_n_; } while (0)
#else
#define ESMI_CT_INIT(slot) do {} while (0)
#define ESMI_CT() do {} while (0)
#endif

#ifndef ESMI_CHAIN_WPS
#define ESMI_CHAIN_WPS 1   // __launch_bounds__ waves/SIMD of the one-wave chain kernels (3 => at most 168 VGPRs)
#endif

namespace esmi {

// acc[nt] += sum over taps j < ntaps of  A_j(32 x K) * W_j[n0 + 32nt + (0..31)][wcol0 + (0..K-1)]^T
//   a_rows[j] : this lane's A row of tap j, + 4*h  (LDS or global), or nullptr for an all-zero row
//   W_j       : W + j*w_tap_stride, row-major (n, ldw); rows >= n_valid contribute zeros
// Operands are fetched in groups of four k-steps (one memory round trip per 32 channels) and the groups of
// ALL taps form one software pipeline: group f+1 is in flight while the 16*NT MFMAs of group f execute.
// (Without the pipeline every group exposed a full L2 latency behind its dependent MFMA chain.)
template <int NT, int MAXTAPS>
__device__ __forceinline__ void wave_gemm_taps(f32x16 (&acc)[NT], const float* const (&a_rows)[MAXTAPS], int ntaps, int K,
                                               const float* __restrict__ W, long w_tap_stride, int ldw, int wcol0, int n0,
                                               int n_valid, int lane) {
    const int i = lane & 31, h = lane >> 5;
    long wofs[NT];
    bool wok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + 32 * nt + i;
        wok[nt] = n < n_valid;
        wofs[nt] = (long)(wok[nt] ? n : 0) * ldw + wcol0 + 4 * h;
    }
    const int ng = K >> 5;            // K is a multiple of 32 on every call site
    const int total = ntaps * ng;
    struct Grp { f32x4 a[4]; f32x4 b[4][NT]; };
    auto fetch = [&](int f, Grp& gq) __attribute__((always_inline)) {
        int j = 0, g = f;
#pragma unroll
        for (int t = 1; t < MAXTAPS; ++t)
            if (g >= ng && t < ntaps) { g -= ng; j = t; }
        const float* ar = a_rows[0];
#pragma unroll
        for (int t = 1; t < MAXTAPS; ++t)
            if (j == t) ar = a_rows[t];
        const float* wj = W + (long)j * w_tap_stride;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            gq.a[q] = ar ? ld4(ar + 32 * g + 8 * q) : zero4();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) gq.b[q][nt] = wok[nt] ? ld4(wj + wofs[nt] + 32 * g + 8 * q) : zero4();
        }
    };
    auto mma = [&](const Grp& gq) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32(gq.a[q][s], gq.b[q][nt][s], acc[nt]);
            }
        }
    };
    Grp g0, g1;
    fetch(0, g0);
    int f = 0;
    for (; f + 2 <= total; f += 2) {   // two groups per trip: the buffers alternate without register copies
        fetch(f + 1, g1);
        mma(g0);
        if (f + 2 < total) fetch(f + 2, g0);
        mma(g1);
    }
    if (f < total) mma(g0);
}

template <int NT>
__device__ __forceinline__ void wave_gemm(f32x16 (&acc)[NT], const float* a_row, int K, const float* __restrict__ W,
                                          int ldw, int wcol0, int n0, int n_valid, int lane) {
    const float* const rows[1] = {a_row};
    wave_gemm_taps<NT, 1>(acc, rows, 1, K, W, 0, ldw, wcol0, n0, n_valid, lane);
}

// C/D-layout accumulators -> LDS tile rows [0,32): tile[row][col0 + 32nt + i]
//
// The row stride is made OPAQUE to the optimiser on purpose.  With a compile-time stride hipcc (ROCm 7.2, gfx950)
// merged two of these ds_write_b32 (rows at +68 and +136 dwords, accumulators in AGPRs) into
// `ds_write2_b32 ... offset0:17 offset1:136` in enc_merge_qkv_kernel<*,2> -- the first offset scaled by 4
// twice -- which put row 1 inside row 0 (caught by the GPU parity tests; the CPU wave simulator, which does not
// go through this backend, was right).  Without constant offsets there is nothing to merge; volatile stores
// also avoid it but serialise against the weight prefetches (+75 % on enc_fuse_va_kernel).
template <int NT>
__device__ __forceinline__ void tile_store(float* tile, int ld, int col0, const f32x16 (&v)[NT], int lane) {
    const int i = lane & 31;
#ifndef ESMI_WAVESIM
    asm volatile("" : "+v"(ld));
#endif
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[tile_row(r, lane) * ld + col0 + 32 * nt + i] = v[nt][r];
    }
}

template <int NT>
__device__ __forceinline__ void zero_tiles(f32x16 (&v)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) v[nt] = zero16();
}

}  // namespace esmi
