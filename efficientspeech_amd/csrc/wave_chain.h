// wave_chain -- building blocks for the fused encoder-side kernels.
//
// A "chain" kernel gives ONE wave (a 64-thread workgroup) a 32-row tile of an utterance and carries it
// through several dependent GEMMs without leaving the CU: the MFMA C/D accumulators of one stage are
// written to a private LDS tile [rows][K+4] and re-read as the next stage's A fragments (lane
// (i = lane&31, h = lane>>5) reads 16 bytes: channels [8kc+4h, +4) of row i).  Convolutions along the
// sequence are row shifts of the A fragment inside the tile (halo rows are recomputed by neighbouring
// tiles); weights stream from L2 in pre-packed MFMA-fragment order, four k-steps per round trip.
//
// Rule (profiles/r05_probes/fuse_va_wrong_rows.md): a loop of ds_swizzle round trips that follows a loop of DPP reductions gets a
// sched_fence() between the two -- with ROCm 7.2 the machine scheduler's interleaving of the two stages over batches of >= 8 rows gave
// wrong statistics on the GPU (right on the simulator); the row-by-row forms in esmi_dev.h are what the fallback kernels use.
#pragma once
#include "esmi_dev.h"

#ifdef ESMI_CHAIN_TRACE
#ifdef ESMI_CT_REALTIME
#define ESMI_CT_CLOCK __builtin_amdgcn_s_memrealtime   // constant 100 MHz
#else
#define ESMI_CT_CLOCK __builtin_amdgcn_s_memtime       // shader clock
#endif
#ifndef ESMI_CT_BLOCK
#define ESMI_CT_BLOCK 7
#define ESMI_CT_THREAD 0
#endif
// development only: shader-clock stamps of workgroup 7 of each chain kernel -> g_chain_trace[kernel_slot*64 + n]
static __device__ long long* g_chain_trace_dev = nullptr;   // one per translation unit (ESMI_TU_CHAIN_TRACE_SETTER, launch.h)
#define ESMI_CT_INIT(slot) int ct_n_ = 0; const bool ct_on_ = blockIdx.x == ESMI_CT_BLOCK && threadIdx.x == ESMI_CT_THREAD; const int ct_slot_ = (slot)
#define ESMI_CT() do { if (ct_on_ && g_chain_trace_dev) g_chain_trace_dev[ct_slot_ * 64 + ct_n_] = (long long)ESMI_CT_CLOCK(); ++ct_n_; } while (0)
#else
#define ESMI_CT_INIT(slot) do {} while (0)
#define ESMI_CT() do {} while (0)
#endif

namespace esmi {
constexpr int kChainRingNt1 = 6, kChainRingNt2 = 4;   // operand-ring depth (groups in flight) of GEMMs one / two column tiles wide (4 .. 10: +-1 %)
constexpr int kChainWps = 1;                           // __launch_bounds__ waves per SIMD of the 32-row-tile chain kernels
}

namespace esmi {

// Weights of every GEMM in the chain kernels are pre-packed once per checkpoint in MFMA B-fragment order
// (pack_bfrag_kernel, small_kernels.h): for a row-major (N, K) matrix, NTW = ceil(N / 32),
//     Wp[((kc*NTW + nt)*64 + lane)*4 + s] = W[32nt + (lane&31)][8kc + 4(lane>>5) + s]      (0 for rows >= N)
// so one wave-level operand fetch (four k-steps of one 32-column tile) is ONE fully coalesced 1 KiB
// global_load_dwordx4.  (Round-1 finding: reading row-major weights -- 32 rows x 32 B per instruction, every
// 128-byte line touched by four different instructions -- made the L1/L2 path, not the MFMA pipe, the limit of
// these kernels: enc_merge_qkv spent 100k cycles on a 49k-cycle MFMA chain.)
//
// acc[nt] += sum over the MAXTAPS taps j of  A_j(32 x 32*KG) * W_j[32(nt0+nt) + (0..31)][8kc0 + (0..32*KG-1)]^T
//   a_rows[j] : this lane's A row of tap j, + 4*h  (LDS or global; must be a readable address even when masked)
//   a_ok[j]   : false -> this lane's row of tap j is all zero (MASKED = false: no row is ever masked)
//   W_j       : Wp + j*w_tap_stride, packed as above with NTW = ntw column tiles; tiles >= ntw contribute zeros
// Operands are fetched in groups of four k-steps (one memory round trip per 32 channels) and the groups of
// ALL taps form one software pipeline: group n+1 is in flight while the 16*NT MFMAs of group n execute.
// Tap COUNT, tap indices and buffer parity are compile-time everywhere (a run-time tap count puts a branch between the
// steps, after which hipcc can only wait with vmcnt(0): no prefetch distance left): an earlier version selected the tap's row pointer
// with a runtime index, which hipcc turned into a scratch array of generic pointers + flat_load (vmcnt AND
// lgkmcnt), and the pipeline collapsed to one exposed round trip per group (127 instead of 70 cycles per MFMA).
template <int NT>
__device__ __forceinline__ void zero_tiles(f32x16 (&v)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) v[nt] = zero16();
}

template <int NT>
struct WaveGrp { f32x4 a[4]; f32x4 b[4][NT]; };

// this lane's base into a packed matrix: column tile nt0 (clamped to the last tile: results of tiles >= ntw are
// garbage the caller must drop), k-step kc0
__device__ __forceinline__ const float* wave_wbase(const float* Wp, int ntw, int kc0, int nt0, int lane) {
    return Wp + ((long)kc0 * ntw + nt0) * 256 + 4 * lane;
}

// weights of group g (four k-steps) of NT column tiles; tiles beyond the matrix alias its last tile
template <int NT>
__device__ __forceinline__ void wave_fetch_b(WaveGrp<NT>& gq, const float* wl, int ntw, int nt0, int g) {
    const float* wg = wl + (long)(4 * g) * ntw * 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int ntc = nt0 + nt < ntw ? nt : ntw - 1 - nt0;   // wave-uniform clamp, no branch around the load
            gq.b[q][nt] = ld4(wg + (q * ntw + ntc) * 256);
        }
    }
}

template <int NT, bool MASKED>
__device__ __forceinline__ void wave_fetch_a(WaveGrp<NT>& gq, const float* ar, bool ok, int g) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        gq.a[q] = ld4(ar + 32 * g + 8 * q);
        if (MASKED && !ok) gq.a[q] = zero4();
    }
}

// 16*NT MFMAs of one group.  A dependent v_mfma_f32_32x32x2_f32 (same accumulator) issues only ~114 cycles after its
// predecessor instead of 64 (probe_wavegemm: NT = 1 ran at 114 cycles per MFMA), so narrow GEMMs (NT <= 2) alternate
// between two accumulator sets by k-step; the caller adds them up at the end (wave_acc_join).
template <int NT>
__device__ __forceinline__ void wave_grp_mma(f32x16 (&acc)[NT], f32x16 (&acc2)[NT], const WaveGrp<NT>& gq) {
#if ESMI_CHAIN_SPLIT
    // split-f16x2 (esmi_dev.h): the group's 32 channels are two 16-channel steps; the weights arrive pre-split (slots
    // 2st, 2st + 1 = pieces 1, 2 of step st: pack_bfrag_kernel), the A rows are split here.  3 MFMAs of 8 passes per step
    // and tile instead of 8 MFMAs of 16 passes.
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const f16x2p a2 = split_f16x2(gq.a[2 * st], gq.a[2 * st + 1]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const u32x4 b1 = __builtin_bit_cast(u32x4, gq.b[2 * st][nt]), b2 = __builtin_bit_cast(u32x4, gq.b[2 * st + 1][nt]);
            if (NT <= 2) {      // two accumulator sets: no MFMA waits for its predecessor's result
                if (st == 0) {
                    acc2[nt] = mfma32_f16(a2.h2, b1, acc2[nt]);
                    acc[nt] = mfma32_f16(a2.h1, b2, acc[nt]);
                    acc2[nt] = mfma32_f16(a2.h1, b1, acc2[nt]);
                } else {
                    acc[nt] = mfma32_f16(a2.h2, b1, acc[nt]);
                    acc2[nt] = mfma32_f16(a2.h1, b2, acc2[nt]);
                    acc[nt] = mfma32_f16(a2.h1, b1, acc[nt]);
                }
            } else {
                acc[nt] = mfma32_split2(a2, b1, b2, acc[nt]);
            }
        }
    }
#else
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (NT <= 2 && (s & 1)) acc2[nt] = mfma32(gq.a[q][s], gq.b[q][nt][s], acc2[nt]);
                else acc[nt] = mfma32(gq.a[q][s], gq.b[q][nt][s], acc[nt]);
            }
        }
    }
#endif
}
// End of a GEMM: add the two accumulator sets.  The split path REQUIRES acc to have been zero at the start of the GEMM: its
// weights carry a factor 2^8 that is taken out here (accumulate across GEMMs in a separate tile: enc_attn_ffn's proj).
template <int NT>
__device__ __forceinline__ void wave_acc_join(f32x16 (&acc)[NT], const f32x16 (&acc2)[NT]) {
#if ESMI_CHAIN_SPLIT
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if (NT <= 2) acc[nt] = (acc[nt] + acc2[nt]) * kF16WScaleInv;
        else acc[nt] *= kF16WScaleInv;
    }
#else
    if (NT <= 2) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] += acc2[nt];
    }
#endif
}

// Request the weights of the FIRST group of a GEMM ahead of time (before the tile store / LDS hand-off / reduction that
// precedes the GEMM), so that every stage of a chain starts with its operands already on the way.
template <int NT>
__device__ __forceinline__ void wave_prefetch(WaveGrp<NT>& g0, const float* __restrict__ Wp, int ntw, int kc0, int nt0, int lane) {
    wave_fetch_b<NT>(g0, wave_wbase(Wp, ntw, kc0, nt0 < ntw ? nt0 : ntw - 1, lane), ntw, nt0 < ntw ? nt0 : ntw - 1, 0);
    sched_fence();
}

// g0.b must hold the weights of (tap 0, group 0): wave_prefetch(g0, Wp, ntw, kc0, nt0, lane).
// The pipeline is a ring of D group buffers with compile-time indices; D grows as the GEMM gets narrower, because a
// group of a narrow GEMM is only 16*NT MFMAs (~1k cycles at NT = 1) while an L2 round trip under load is 2-3k.
template <int NT, int MAXTAPS, int KG, bool MASKED>
__device__ __forceinline__ void wave_gemm_taps(f32x16 (&acc)[NT], WaveGrp<NT>& g0, const float* const (&a_rows)[MAXTAPS],
                                               const bool (&a_ok)[MAXTAPS], const float* __restrict__ Wp,
                                               long w_tap_stride, int ntw, int kc0, int nt0, int lane) {
    if (nt0 >= ntw) nt0 = ntw - 1;
    const float* wl = wave_wbase(Wp, ntw, kc0, nt0, lane);
    f32x16 acc2[NT];
    zero_tiles<NT>(acc2);
    wave_fetch_a<NT, MASKED>(g0, a_rows[0], a_ok[0], 0);
    if constexpr (NT >= 4 && KG % 2 == 0 && KG >= 4) {
        // wide and long: taps unrolled, group pairs of one tap in a rolled loop (code size), two buffers
        WaveGrp<NT> g1;
#pragma unroll
        for (int j = 0; j < MAXTAPS; ++j) {
            {
                const float* wj = wl + (long)j * w_tap_stride;
                const int jn = j + 1 < MAXTAPS ? j + 1 : j;
                for (int g = 0; g < KG; g += 2) {
                    wave_fetch_b<NT>(g1, wj, ntw, nt0, g + 1);
                    wave_fetch_a<NT, MASKED>(g1, a_rows[j], a_ok[j], g + 1);
                    sched_fence();
                    wave_grp_mma<NT>(acc, acc2, g0);
                    sched_fence();
                    if (g + 2 < KG) {
                        wave_fetch_b<NT>(g0, wj, ntw, nt0, g + 2);
                        wave_fetch_a<NT, MASKED>(g0, a_rows[j], a_ok[j], g + 2);
                    } else if (j + 1 < MAXTAPS) {
                        wave_fetch_b<NT>(g0, wj + w_tap_stride, ntw, nt0, 0);
                        wave_fetch_a<NT, MASKED>(g0, a_rows[jn], a_ok[jn], 0);
                    }
                    sched_fence();
                    wave_grp_mma<NT>(acc, acc2, g1);
                    sched_fence();
                }
            }
        }
    } else {
        // fully unrolled: step n = j*KG + g lives in ring slot n % D (slot 0 = g0); everything is compile-time
        constexpr int STEPS = MAXTAPS * KG;
        constexpr int D0 = NT == 1 ? kChainRingNt1 : (NT == 2 ? kChainRingNt2 : 2);
        constexpr int D = D0 < STEPS ? D0 : (STEPS > 1 ? STEPS : 2);
        WaveGrp<NT> ring[D - 1];
        auto slot = [&](int n) __attribute__((always_inline)) -> WaveGrp<NT>& { return n % D == 0 ? g0 : ring[n % D - 1]; };
        auto fetch = [&](int m) __attribute__((always_inline)) {   // operands of step m, if that step exists
            const int jm = m / KG < MAXTAPS ? m / KG : MAXTAPS - 1, gm = m % KG;
            if (m < STEPS) {
                wave_fetch_b<NT>(slot(m), wl + (long)jm * w_tap_stride, ntw, nt0, gm);
                wave_fetch_a<NT, MASKED>(slot(m), a_rows[jm], a_ok[jm], gm);
            }
        };
#pragma unroll
        for (int m = 1; m < D - 1; ++m) fetch(m);
#pragma unroll
        for (int n = 0; n < STEPS; ++n) {
            fetch(n + D - 1);
            sched_fence();
            wave_grp_mma<NT>(acc, acc2, slot(n));
            sched_fence();
        }
    }
    wave_acc_join<NT>(acc, acc2);
}

// single-tap GEMM with a compile-time K = 32*KG (deep static pipeline)
template <int NT, int KG>
__device__ __forceinline__ void wave_gemm_k(f32x16 (&acc)[NT], WaveGrp<NT>& g0, const float* a_row, bool ok,
                                            const float* __restrict__ Wp, int ntw, int kc0, int nt0, int lane) {
    const float* const rows[1] = {a_row};
    const bool oks[1] = {ok};
    wave_gemm_taps<NT, 1, KG, true>(acc, g0, rows, oks, Wp, 0, ntw, kc0, nt0, lane);
}

// single-tap GEMM with a run-time K (a multiple of 32); a_row must be readable, ok = false -> zero row.
// g0.b must hold the weights of group 0 (wave_prefetch).
template <int NT>
__device__ __forceinline__ void wave_gemm(f32x16 (&acc)[NT], WaveGrp<NT>& g0, const float* a_row, bool ok, int K,
                                          const float* __restrict__ Wp, int ntw, int kc0, int nt0, int lane) {
    if (nt0 >= ntw) nt0 = ntw - 1;
    const float* wl = wave_wbase(Wp, ntw, kc0, nt0, lane);
    const int ng = K >> 5;
    WaveGrp<NT> g1;
    f32x16 acc2[NT];
    zero_tiles<NT>(acc2);
    wave_fetch_a<NT, true>(g0, a_row, ok, 0);
    int f = 0;
    for (; f + 2 <= ng; f += 2) {   // two groups per trip: the buffers alternate without register copies
        wave_fetch_b<NT>(g1, wl, ntw, nt0, f + 1);
        wave_fetch_a<NT, true>(g1, a_row, ok, f + 1);
        sched_fence();
        wave_grp_mma<NT>(acc, acc2, g0);
        sched_fence();
        if (f + 2 < ng) {
            wave_fetch_b<NT>(g0, wl, ntw, nt0, f + 2);
            wave_fetch_a<NT, true>(g0, a_row, ok, f + 2);
        }
        sched_fence();
        wave_grp_mma<NT>(acc, acc2, g1);
        sched_fence();
    }
    if (f < ng) wave_grp_mma<NT>(acc, acc2, g0);
    wave_acc_join<NT>(acc, acc2);
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
    ld = opaque_i(ld);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[tile_row(r, lane) * ld + col0 + 32 * nt + i] = v[nt][r];
    }
}


}  // namespace esmi
