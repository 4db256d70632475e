// chain16 -- building blocks of the round-5 encoder-side chain kernels (enc_va16.h, enc_block16.h).
//
// What changed against wave_chain.h (one wave = one 32-row tile, weights streamed per wave from L2, accumulators in the C/D layout):
//  * a wave owns a 16-ROW tile and runs v_mfma_f32_16x16x32_f16, so a 128-position sequence is 8 waves = TWO waves per SIMD: the
//    dependent chains of two tiles interleave on every SIMD (a lone wave issues a VALU instruction every ~5 cycles);
//  * every weight byte is fetched ONCE PER WORKGROUP: the stage's packed matrices are copied verbatim from global memory into LDS by
//    LDS-DMA (global_load_lds_dwordx4, no staging registers) one stage ahead, and all waves read their fragments with ds_read_b128
//    (round 4 measured the per-wave L2 streams as the limiter: 4 -> 7 waves per CU took 1.78x as long);
//  * products are computed TRANSPOSED (weights are the first MFMA operand): D^T[channel][row], i.e. lane (i = lane & 15, g = lane >> 4)
//    holds row i and, per 16-channel output tile nt, the four consecutive channels 16 nt + 4 g + (0..3).  Row-wise work (LayerNorm,
//    row dots, softmax, bucketize, masks) is then in-lane plus two lane-group exchanges (v_permlane16_swap, v_permlane32_swap) for
//    all 16 rows at once -- the C/D layout needed a 32-lane DPP / ds_swizzle reduction per row and register -- and every LDS / global
//    access of a result is 8 or 16 bytes wide;
//  * activations that feed a GEMM live in LDS already split into the two binary16 planes (esmi_dev.h): split once where they are
//    produced, not once per tap / consumer.
//
// Operand conventions.  The weights are the arrays esmi_pack_bfrag_f32 makes (small_kernels.h, the split build): per (32-channel
// group G, 32-row tile nt) four 1 KiB slots {step 0 piece 1, step 0 piece 2, step 1 piece 1, step 1 piece 2}; in a slot lane
// (i32, h) holds W[32 nt + i32][32 G + 16 st + 4 h + {0..3, 8..11}].  The A operand of v_mfma_f32_16x16x32_f16 for output tile
// nt16 and K group G is lane (i, g) -> 16 bytes of slot st = g >> 1 at lane 16 (nt16 & 1) + i + 32 (g & 1) of tile nt16 >> 1: the
// lane's eight k-slots are the channels 32 G + 16 (g >> 1) + 4 (g & 1) + {0..3, 8..11}.  The B operand (activations) must present
// the same channels in the same order: fp32 rows in global memory are read as two float4 (channels c0 .. c0 + 3 and c0 + 8 .. c0 + 11)
// and split on the fly; the f16 planes in LDS store each 16-channel step in the order [0..3, 8..11, 4..7, 12..15], so the eight
// k-slots of a lane are 16 contiguous bytes per plane (only the WRITER permutes: planes_store).
#pragma once
#include "esmi_dev.h"
#include "wave_chain.h"

namespace esmi {
namespace c16 {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// (swap16_f -- lane l <- lane l ^ 16, v_permlane16_swap -- lives in wavesim_shim.h with the other target-specific primitives)
// sum over the four lanes that hold one row (lanes i, i + 16, i + 32, i + 48); every lane ends with the same bits
__device__ __forceinline__ float row_sum4(float v) {
    v += swap16_f(v);
    v += swap32_f(v);
    return v;
}
__device__ __forceinline__ float row_max4(float v) {
    v = fmaxf(v, swap16_f(v));
    v = fmaxf(v, swap32_f(v));
    return v;
}

// D^T[16 channels][16 rows] += W(16 x 32) . A^T(32 x 16): the three significant products of the fp32-accurate split (esmi_dev.h);
// the caller takes the weights' 2^8 out again (kF16WScaleInv)
__device__ __forceinline__ f32x4 mma(const u32x4& w1, const u32x4& w2, const f16x2p& a, f32x4 c) {
    c = mfma16_f16(w1, a.h2, c);
    c = mfma16_f16(w2, a.h1, c);
    c = mfma16_f16(w1, a.h1, c);
    return c;
}

// per-lane part of a weight-fragment address (floats) for matrices of NTW 32-row tiles: see the header comment
__device__ __forceinline__ int wlane(int lane, int NTW) {
    const int i = lane & 15, g = lane >> 4;
    return (g >> 1) * (2 * NTW * 256) + 4 * (i + 32 * (g & 1));
}
// acc[nt] += W[16 nt + (0..15)][32 (G0 + ks) + (0..31)] . act^T for ks < KS, nt < NT16: W0 = the packed (32 NTW x K) matrix in LDS,
// lw = wlane(lane, NTW), bop(ks) = this lane's B operand of k-step ks
template <int NT16, int NTW, int KS, typename BOp>
__device__ __forceinline__ void gemm(f32x4 (&acc)[NT16], const float* W0, int lw, int G0, BOp bop) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const f16x2p a = bop(ks);
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt) {
            const float* wp = W0 + ((4 * (G0 + ks)) * NTW + (nt >> 1)) * 256 + 64 * (nt & 1) + lw;
            const u32x4 w1 = *reinterpret_cast<const u32x4*>(wp), w2 = *reinterpret_cast<const u32x4*>(wp + NTW * 256);
            acc[nt] = mma(w1, w2, a, acc[nt]);
        }
    }
}

// The same with every weight fragment of a k-step requested before the step's first product (one LDS round trip per step instead of
// one per output tile -- hipcc emits `ds_read, ds_read, s_waitcnt lgkmcnt(0), 3 x v_mfma` per tile for gemm() above) and the products
// issued product-major, so that consecutive MFMAs write different accumulators.
template <int N>
struct WFrags { u32x4 w1[N], w2[N]; };
template <int NT16, int NTW, int N>
__device__ __forceinline__ void wfrags_load(WFrags<N>& f, int at, const float* W0, int lw, int G) {
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt) {
        const float* wp = W0 + ((4 * G) * NTW + (nt >> 1)) * 256 + 64 * (nt & 1) + lw;
        f.w1[at + nt] = *reinterpret_cast<const u32x4*>(wp);
        f.w2[at + nt] = *reinterpret_cast<const u32x4*>(wp + NTW * 256);
    }
}
template <int N>
__device__ __forceinline__ void mma_all(f32x4 (&acc)[N], const WFrags<N>& f, const f16x2p& a) {
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = mfma16_f16(f.w1[n], a.h2, acc[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = mfma16_f16(f.w2[n], a.h1, acc[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = mfma16_f16(f.w1[n], a.h1, acc[n]);
}
// small GEMM, all fragments up front: acc[nt] += W[16 nt + ..][32 (G0 + ks) + ..] . act^T, ks < KS
template <int NT16, int NTW, int KS, typename BOp>
__device__ __forceinline__ void gemm_pf(f32x4 (&acc)[NT16], const float* W0, int lw, int G0, BOp bop) {
    WFrags<NT16> f[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wfrags_load<NT16, NTW, NT16>(f[ks], 0, W0, lw, G0 + ks);
    sched_fence();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) mma_all<NT16>(acc, f[ks], bop(ks));
}

// ---- activation planes in LDS: row = [C halves piece 1 | C halves piece 2 | pad], row stride LD = C + 4 dwords (rows 16 bytes
// apart mod 256: the 16-byte fragment reads and 8-byte tile writes of 16 consecutive rows are bank-conflict free)
// B operand of k-step ks of a row: rowp = tile + row * LD + 4 g (dwords), chalf = C / 2
__device__ __forceinline__ f16x2p planes_load(const unsigned* rowp, int ks, int chalf) {
    f16x2p a;
    a.h1 = *reinterpret_cast<const u32x4*>(rowp + 16 * ks);
    a.h2 = *reinterpret_cast<const u32x4*>(rowp + 16 * ks + chalf);
    return a;
}
// the lane's four channels 16 nt + 4 g + (0..3) of its row -> both planes: rowp = tile + row * LD + wpos(lane) (dwords)
__device__ __forceinline__ int wpos(int lane) { const int g = lane >> 4; return 4 * (g & 1) + 2 * (g >> 1); }
__device__ __forceinline__ void planes_store(unsigned* rowp, int nt, int chalf, const f32x4& v) {
    unsigned h1a, h2a, h1b, h2b;
    split_f16_pair(v[0], v[1], h1a, h2a);
    split_f16_pair(v[2], v[3], h1b, h2b);
    *reinterpret_cast<u32x2*>(rowp + 8 * nt) = u32x2{h1a, h1b};
    *reinterpret_cast<u32x2*>(rowp + 8 * nt + chalf) = u32x2{h2a, h2b};
}
// B operand of k-step ks from an fp32 row in global memory: off = byte offset of the row + gl_lane(lane) (kBufOOB: a zero row)
__device__ __forceinline__ unsigned gl_lane(int lane) { const int g = lane >> 4; return (unsigned)((16 * (g >> 1) + 4 * (g & 1)) * 4); }
__device__ __forceinline__ f16x2p global_bop(BufRsrc r, unsigned off, int ks) {
    const f32x4 x0 = buf_ld4(r, off + 128u * ks), x1 = buf_ld4(r, off + 128u * ks + 32u);
    return split_f16x2(x0, x1);
}

// LayerNorm over the 16 NT16 channels of each row, D^T layout (two-pass, biased variance, eps inside the root: nn.LayerNorm)
template <int NT16>
__device__ __forceinline__ void layernorm(f32x4 (&v)[NT16], const f32x4 (&g)[NT16], const f32x4 (&b)[NT16], float eps = 1e-5f) {
    const float inv_c = 1.0f / (float)(16 * NT16);
    float s = 0.0f;
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt) s += (v[nt][0] + v[nt][1]) + (v[nt][2] + v[nt][3]);
    const float mean = row_sum4(s) * inv_c;
    float q = 0.0f;
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[nt][e] -= mean;
            q = fmaf(v[nt][e], v[nt][e], q);
        }
    }
    const float rstd = rsqrt_fast_f32(row_sum4(q) * inv_c + eps);
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[nt][e] = fmaf(v[nt][e] * rstd, g[nt][e], b[nt][e]);
    }
}

// copy `nfrag` KiB-fragments from global memory to LDS verbatim, dealt round-robin to the nw waves of the workgroup (LDS-DMA: no
// registers; complete after the issuing wave's vmcnt drains -- the barrier in front of the first reader does that)
// `rot` rotates the ORDER in which the fragments are requested (the copy itself is unchanged): every workgroup of the grid pulls the same
// bytes at the same moment, and in lockstep order they would all queue on the same L2 channel line after line; a per-workgroup rotation
// spreads the requests over the channels
__device__ __forceinline__ void dma_frags(const float* gsrc, float* ldst, int nfrag, int w, int nw, int lane, int rot = 0) {
    rot %= nfrag;
    for (int f = w; f < nfrag; f += nw) {
        int fr = f + rot;
        fr = fr >= nfrag ? fr - nfrag : fr;
        lds_dma16(gsrc + fr * 256 + 4 * lane, ldst + fr * 256, lane);
    }
}

__device__ __forceinline__ f32x4 ld4_lds(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 fmaf4(const f32x4& v, float s, const f32x4& b) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaf(v[e], s, b[e]);
    return o;
}
__device__ __forceinline__ f32x4 relu4(const f32x4& v) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaxf(v[e], 0.0f);
    return o;
}

}  // namespace c16
}  // namespace esmi
