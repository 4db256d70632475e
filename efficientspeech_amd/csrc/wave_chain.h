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

namespace esmi {

// acc[nt] += A(32 x K) * W[n0 + 32nt + (0..31)][wcol0 + (0..K-1)]^T
//   a_row : this lane's A row + 4*h  (LDS or global), or nullptr for an all-zero row
//   W     : row-major (n, ldw); rows >= n_valid contribute zeros
template <int NT>
__device__ __forceinline__ void wave_gemm(f32x16 (&acc)[NT], const float* a_row, int K, const float* __restrict__ W,
                                          int ldw, int wcol0, int n0, int n_valid, int lane) {
    const int i = lane & 31, h = lane >> 5;
    const float* wrow[NT];
    bool wok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + 32 * nt + i;
        wok[nt] = n < n_valid;
        wrow[nt] = W + (long)(wok[nt] ? n : 0) * ldw + wcol0 + 4 * h;
    }
    for (int kc = 0; kc < (K >> 3); kc += 4) {   // K is a multiple of 32 on every call site
        f32x4 av[4], bv[4][NT];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            av[g] = a_row ? ld4(a_row + 8 * (kc + g)) : zero4();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[g][nt] = wok[nt] ? ld4(wrow[nt] + 8 * (kc + g)) : zero4();
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32(av[g][s], bv[g][nt][s], acc[nt]);
            }
        }
    }
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
