// Device-side primitives shared by the esmi kernels (gfx950 / CDNA4, wave64).
//
// The kernels are written once, against the handful of primitives below.  The product build
// (hipcc --offload-arch=gfx950) maps them to the CDNA4 instructions; the test-only wave
// simulator (tools/wavesim, -DESMI_WAVESIM, host clang++) maps them to lane-accurate CPU
// emulation so that index math / MFMA layouts / LDS protocols can be checked without a GPU.
#pragma once

#ifdef ESMI_WAVESIM
#include "wavesim.h"
#define ESMI_DYN_LDS(name) float* name = (float*)wavesim::dyn_lds()
#define ESMI_LAUNCH(kern, grid, block, lds, stream, ...) \
    wavesim::launch(grid, block, lds, [&]() { kern(__VA_ARGS__); })
#else
#include <hip/hip_runtime.h>
#define ESMI_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) float name[]
#define ESMI_LAUNCH(kern, grid, block, lds, stream, ...) hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__)
#endif

namespace esmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- MFMA, exact fp32 (v_mfma_f32_32x32x2_f32: 64 cycles/SIMD, == k-ordered fmaf chain)
//   A[i = lane&31][k = lane>>5],  B[k = lane>>5][j = lane&31]
//   D reg r of lane l: row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
#ifdef ESMI_WAVESIM
    return wavesim::mfma_32x32x2(a, b, c);
#else
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}

// ---- fp32-accurate products on the bf16 matrix pipe (16x the fp32 MFMA rate on gfx950).
// An fp32 value is split EXACTLY into three bf16 pieces by truncation: x = hi + mid + lo, 8 + 8 + 8 mantissa bits.  The
// product of two bf16 values is exact in fp32, so  a.b ~= hi.hi + hi.mid + mid.hi + mid.mid + hi.lo + lo.hi  (the dropped
// terms are below 2^-24 relative) accumulated in fp32 by v_mfma_f32_32x32x16_bf16 is as accurate as an fp32 FMA chain:
// measured on K = 128 dot products, max error 3.0e-6 vs 8.5e-6 for sequential fp32 accumulation.
//   A[i = lane&31][k = 8*(lane>>5) + (0..7)],  B[k = 8*(lane>>5) + (0..7)][j = lane&31]   (8 bf16 = 4 dwords per lane)
//   D as for the 32x32x2 fp32 MFMA
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct bf16x3 { u32x4 hi, mid, lo; };
__device__ __forceinline__ unsigned pack_hi16(unsigned even, unsigned odd) {   // {odd[31:16], even[31:16]}
#ifdef ESMI_WAVESIM
    return (odd & 0xFFFF0000u) | (even >> 16);
#else
    return __builtin_amdgcn_perm(odd, even, 0x07060302);
#endif
}
__device__ __forceinline__ bf16x3 split_bf16x3(const f32x4& x0, const f32x4& x1) {   // 8 consecutive k of one row
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = j < 4 ? x0[j & 3] : x1[j & 3];
        h[j] = __builtin_bit_cast(unsigned, x) & 0xFFFF0000u;
        const float r1 = x - __builtin_bit_cast(float, h[j]);                  // exact
        m[j] = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
        const float r2 = r1 - __builtin_bit_cast(float, m[j]);                 // exact
        l[j] = __builtin_bit_cast(unsigned, r2);                               // truncated to bf16 by the pack
    }
    bf16x3 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o.hi[j] = pack_hi16(h[2 * j], h[2 * j + 1]);
        o.mid[j] = pack_hi16(m[2 * j], m[2 * j + 1]);
        o.lo[j] = pack_hi16(l[2 * j], l[2 * j + 1]);
    }
    return o;
}
__device__ __forceinline__ f32x16 mfma32_bf16(const u32x4& a, const u32x4& b, f32x16 c) {
#ifdef ESMI_WAVESIM
    return wavesim::mfma_32x32x16_bf16(a, b, c);
#else
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#endif
}
// acc += A(32 x 16, fp32 split on the fly) . B(16 x 32, pre-split planes)
__device__ __forceinline__ f32x16 mfma32_split(const bf16x3& a, const u32x4& bh, const u32x4& bm, const u32x4& bl, f32x16 c) {
    c = mfma32_bf16(a.hi, bh, c);
    c = mfma32_bf16(a.hi, bm, c);
    c = mfma32_bf16(a.mid, bh, c);
    c = mfma32_bf16(a.mid, bm, c);
    c = mfma32_bf16(a.hi, bl, c);
    c = mfma32_bf16(a.lo, bh, c);
    return c;
}
// the transposed product: acc(32 out-channels x 32 rows) += W planes . A^T (operands swapped: same fragments, D^T)
__device__ __forceinline__ f32x16 mfma32_split_wx(const u32x4& bh, const u32x4& bm, const u32x4& bl, const bf16x3& a, f32x16 c) {
    c = mfma32_bf16(bh, a.hi, c);
    c = mfma32_bf16(bm, a.hi, c);
    c = mfma32_bf16(bh, a.mid, c);
    c = mfma32_bf16(bm, a.mid, c);
    c = mfma32_bf16(bl, a.hi, c);
    c = mfma32_bf16(bh, a.lo, c);
    return c;
}

// ---- the same idea on the f16 matrix pipe with HALF the products: x = h1 + h2 (two binary16 pieces, 11 + 11 significand
// bits; h1 by round-toward-zero so the residual x - h1 is exact), weights pre-scaled by 2^8 and pre-split the same way
// (round to nearest), a.w ~= h1.w1 + h1.w2 + h2.w1: the dropped h2.w2 term and the pieces' rounding are ~2^-22 relative,
// below the fp32 accumulation error of a K >= 32 contraction (numpy emulation, K = 128: max error 0.8-3.2e-6 vs 1.4-4.6e-6
// for sequential fp32 accumulation).  Operands must be inside the binary16 range: |a| < 65504 and
// |w| < 255 (2^8 scale keeps the second piece of weights down to 5e-4 a normal number; smaller ones lose nothing that
// matters: absolute error < 2.4e-10 per weight).  Layout as for v_mfma_f32_32x32x16_bf16.
// ---- activation-range check (the `libesmi_checked.so` build, -DESMI_RANGE_CHECK=1): the split-f16 contractions need their operands
// inside the binary16 range (|a| < 65504; the first piece saturates there, so larger values are silently wrong, not inf).  Trained
// networks are orders of magnitude inside it (LayerNorm / tanh / GELU outputs, O(1) embeddings), so the product kernels do not pay
// for a test; the checked build ORs a device word whenever a value entering a split is out of range, and
// esmi_phoneme2mel_forward_f32 turns that into ESMI_ERR_RANGE.  One flag pointer per translation unit (no relocatable device code).
#ifndef ESMI_RANGE_CHECK
#define ESMI_RANGE_CHECK 0
#endif
#if ESMI_RANGE_CHECK
#ifdef ESMI_WAVESIM
static int* g_esmi_range_flag = nullptr;
#else
static __device__ int* g_esmi_range_flag = nullptr;
#endif
__device__ __forceinline__ void range_note(float a) {
    if (!(fabsf(a) < 65504.0f)) {       // also true for nan
        int* f = g_esmi_range_flag;
        if (f) *f = 1;                  // (a plain store of the same value from any number of lanes: no atomic needed)
    }
}
#else
__device__ __forceinline__ void range_note(float) {}
#endif

struct f16x2p { u32x4 h1, h2; };
#ifndef ESMI_CHAIN_SPLIT
#define ESMI_CHAIN_SPLIT 1   // weight GEMMs of the encoder-side chain kernels: 1 = split-f16x2 (3 f16 MFMAs per 16 channels), 0 = fp32 MFMA
#endif
constexpr float kF16WScale = 256.0f, kF16WScaleInv = 1.0f / 256.0f;
__host__ __device__ inline unsigned f32_to_f16_bits(float f, bool rtz) {     // software conversion (packers, simulator)
    const unsigned u = __builtin_bit_cast(unsigned, f), sign = (u >> 16) & 0x8000u, a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return sign | 0x7C00u | (a > 0x7F800000u ? 0x200u : 0u);
    const int e = (int)(a >> 23) - 127;
    if (a == 0 || e < -26) return sign;
    if (e > 15) return sign | (rtz ? 0x7BFFu : 0x7C00u);
    const unsigned m = (a & 0x7FFFFFu) | 0x800000u;
    const int shift = e >= -14 ? 13 : 13 + (-14 - e);
    if (shift > 25) return sign;
    unsigned q = m >> shift;
    const unsigned rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (!rtz && (rem > half || (rem == half && (q & 1u)))) ++q;
    unsigned h = e >= -14 ? ((unsigned)(e + 14) << 10) + q : q;
    if (rtz && h >= 0x7C00u) h = 0x7BFFu;
    return sign | h;
}
__host__ __device__ inline float f16_bits_to_f32(unsigned h) {
    const unsigned sign = (h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3FFu;
    if (e == 31u) return __builtin_bit_cast(float, sign | 0x7F800000u | (m << 13));
    if (e == 0u) {   // zero / subnormal: m * 2^-24
        const float v = (float)m * 5.9604644775390625e-08f;
        return (h & 0x8000u) ? -v : v;
    }
    return __builtin_bit_cast(float, sign | ((e + 112u) << 23) | (m << 13));
}
__device__ __forceinline__ void split_f16_pair(float a, float b, unsigned& h1, unsigned& h2) {   // {b, a} pieces, a in the low half
    range_note(a);
    range_note(b);
#ifdef ESMI_WAVESIM
    const unsigned ha = f32_to_f16_bits(a, true), hb = f32_to_f16_bits(b, true);
    const float ra = a - f16_bits_to_f32(ha), rb = b - f16_bits_to_f32(hb);
    h1 = ha | (hb << 16);
    h2 = f32_to_f16_bits(ra, true) | (f32_to_f16_bits(rb, true) << 16);
#else
    const auto h = __builtin_amdgcn_cvt_pkrtz(a, b);            // v_cvt_pkrtz_f16_f32
    h1 = __builtin_bit_cast(unsigned, h);
    // (v_fma_mix_f32 would fold the conversion into the subtraction -- tried as inline asm in round 2: no measurable gain in
    // the decoder, and enc_fuse_va mis-computed one row of the B = 1 fox fixture with it, so the plain form stays)
    const float ra = a - (float)h[0], rb = b - (float)h[1];     // exact
    h2 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
#endif
}
// the same with both pieces rounded to nearest (what the weight packers do): one bit more than the truncating form, three
// conversions instead of one packed one -- used where a value is split once and read many times (weight staging)
__device__ __forceinline__ void split_f16_pair_rn(float a, float b, unsigned& h1, unsigned& h2) {
    range_note(a);
    range_note(b);
#ifdef ESMI_WAVESIM
    const unsigned ha = f32_to_f16_bits(a, false), hb = f32_to_f16_bits(b, false);
    const float ra = a - f16_bits_to_f32(ha), rb = b - f16_bits_to_f32(hb);
    h1 = ha | (hb << 16);
    h2 = f32_to_f16_bits(ra, false) | (f32_to_f16_bits(rb, false) << 16);
#else
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    const f16x2_t h = {(_Float16)a, (_Float16)b};                 // v_cvt_f16_f32: round to nearest even
    const float ra = a - (float)h[0], rb = b - (float)h[1];       // exact
    const f16x2_t r = {(_Float16)ra, (_Float16)rb};
    h1 = __builtin_bit_cast(unsigned, h);
    h2 = __builtin_bit_cast(unsigned, r);
#endif
}
__device__ __forceinline__ f16x2p split_f16x2(const f32x4& x0, const f32x4& x1) {   // 8 consecutive k of one row
    f16x2p o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = j < 2 ? x0[2 * j] : x1[2 * j - 4], b = j < 2 ? x0[2 * j + 1] : x1[2 * j - 3];
        unsigned h1, h2;
        split_f16_pair(a, b, h1, h2);
        o.h1[j] = h1;
        o.h2[j] = h2;
    }
    return o;
}
// 8 consecutive k of one row rounded to binary16 (nearest even), in the k-slot order of split_f16x2: the single-piece operand of the
// `precision=16` training GEMMs
__device__ __forceinline__ u32x4 round_f16x8(const f32x4& x0, const f32x4& x1) {
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = j < 2 ? x0[2 * j] : x1[2 * j - 4], b = j < 2 ? x0[2 * j + 1] : x1[2 * j - 3];
        range_note(a);
        range_note(b);
#ifdef ESMI_WAVESIM
        o[j] = f32_to_f16_bits(a, false) | (f32_to_f16_bits(b, false) << 16);
#else
        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
        const f16x2_t h = {(_Float16)a, (_Float16)b};
        o[j] = __builtin_bit_cast(unsigned, h);
#endif
    }
    return o;
}
__device__ __forceinline__ f32x16 mfma32_f16(const u32x4& a, const u32x4& b, f32x16 c) {
#ifdef ESMI_WAVESIM
    return wavesim::mfma_32x32x16_f16(a, b, c);
#else
    typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
#endif
}
// v_mfma_f32_16x16x32_f16 (gfx950): A[i = lane&15][k = 8*(lane>>4) + (0..7)], B[k][j = lane&15]; D reg r: row 4*(lane>>4) + r, col lane&15
__device__ __forceinline__ f32x4 mfma16_f16(const u32x4& a, const u32x4& b, f32x4 c) {
#ifdef ESMI_WAVESIM
    return wavesim::mfma_16x16x32_f16(a, b, c);
#else
    typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
#endif
}
// acc += A(32 x 16, fp32 split on the fly) . B(16 x 32, pre-scaled pre-split planes); the caller rescales by kF16WScaleInv
__device__ __forceinline__ f32x16 mfma32_split2(const f16x2p& a, const u32x4& b1, const u32x4& b2, f32x16 c) {
    c = mfma32_f16(a.h2, b1, c);
    c = mfma32_f16(a.h1, b2, c);
    c = mfma32_f16(a.h1, b1, c);
    return c;
}
__device__ __forceinline__ f32x16 mfma32_split2_wx(const u32x4& b1, const u32x4& b2, const f16x2p& a, f32x16 c) {   // D^T
    c = mfma32_f16(b1, a.h2, c);
    c = mfma32_f16(b2, a.h1, c);
    c = mfma32_f16(b1, a.h1, c);
    return c;
}

__device__ __forceinline__ float shfl_xor_f(float v, int mask) {
#ifdef ESMI_WAVESIM
    return wavesim::shfl_xor(v, mask);
#else
    return __shfl_xor(v, mask, 64);
#endif
}
__device__ __forceinline__ int shfl_up_i(int v, int delta) {
#ifdef ESMI_WAVESIM
    return wavesim::shfl_up_i(v, delta);
#else
    return __shfl_up(v, delta, 64);
#endif
}
__device__ __forceinline__ int shfl_i(int v, int src) {
#ifdef ESMI_WAVESIM
    return wavesim::shfl_i(v, src);
#else
    return __shfl(v, src, 64);
#endif
}

__device__ __forceinline__ unsigned long long ballot64(bool pred) {
#ifdef ESMI_WAVESIM
    return wavesim::ballot(pred);
#else
    return __ballot(pred);
#endif
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int lane_id_raw() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// row of accumulator register r inside a 32-row MFMA tile
__device__ __forceinline__ int tile_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---- cross-lane moves inside a 32-lane half wave without an LDS round trip: DPP (VALU rate) for the four
// in-row steps and ds_swizzle SWAPX16 for the row pair.  A ds_bpermute butterfly (what __shfl_xor lowers to)
// costs ~100+ cycles of latency per step; LayerNorm / softmax reductions run 10 steps per tile row.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {   // CTRL: 0xB1 quad xor1, 0x4E quad xor2, 0x141 half mirror, 0x140 mirror
#ifdef ESMI_WAVESIM
    const int l = lane_id_raw();
    int src = l;
    if (CTRL == 0xB1) src = l ^ 1;
    else if (CTRL == 0x4E) src = l ^ 2;
    else if (CTRL == 0x141) src = (l & ~7) | (7 - (l & 7));
    else if (CTRL == 0x140) src = (l & ~15) | (15 - (l & 15));
    return wavesim::shfl(v, src);
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
#endif
}
__device__ __forceinline__ float swz_xor16_f(float v) {   // lane l <- lane l^16 (within each 32-lane half)
#ifdef ESMI_WAVESIM
    return wavesim::shfl(v, lane_id_raw() ^ 16);
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
#endif
}

__device__ __forceinline__ float swap32_f(float v) {     // lane l <- lane l^32, VALU only (v_permlane32_swap, gfx950)
#ifdef ESMI_WAVESIM
    return wavesim::shfl(v, lane_id_raw() ^ 32);
#else
    const unsigned x = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);   // r[0]: lanes 32-63 <- x[0-31]; r[1]: lanes 0-31 <- x[32-63]
    return __builtin_bit_cast(float, lane_id_raw() < 32 ? r[1] : r[0]);
#endif
}

// sum over the 32 lanes that hold one tile row (lanes sharing lane>>5); every lane ends with the same bits
__device__ __forceinline__ float row_sum32(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    v += swz_xor16_f(v);
    return v;
}
__device__ __forceinline__ float row_sum16(float v) {   // all-reduce over each aligned group of 16 lanes (one DPP row)
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    return v;
}
template <int N>   // all-reduce over aligned groups of N = 4, 8 or 16 adjacent lanes
__device__ __forceinline__ float row_sum_n(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    if (N >= 8) v += dpp_f<0x141>(v);
    if (N >= 16) v += dpp_f<0x140>(v);
    return v;
}
__device__ __forceinline__ float row_max32(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    v = fmaxf(v, swz_xor16_f(v));
    return v;
}

// ---- activations (fp32; |err| <= ~2e-7 absolute, far inside the 1e-4 parity budget)
__device__ __forceinline__ float tanh_f32(float x) {
    const float ax = fabsf(x);
    const float e = expf(-2.0f * ax);
    const float t = (1.0f - e) / (1.0f + e);
    return copysignf(t, x);
}
// hardware-transcendental form: 1 - 2/(1 + 2^(2x*log2 e)); v_exp_f32 + v_rcp_f32, |err| ~ 1e-6 absolute
__device__ __forceinline__ float tanh_fast_f32(float x) {
#ifdef ESMI_WAVESIM
    return tanh_f32(x);
#else
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);   // e^(2x); inf for large x -> rcp = 0 -> 1
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
#endif
}
// tanh(s*x + b) with the scale of the exponent folded into the fma: pass s2 = s * 2 log2(e), b2 = b * 2 log2(e)
constexpr float kTanhExpScale = 2.885390081777927f;
__device__ __forceinline__ float tanh_fast_fma_f32(float x, float s2, float b2) {
#ifdef ESMI_WAVESIM
    return tanh_f32(fmaf(x, s2, b2) * (1.0f / kTanhExpScale));
#else
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(x, s2, b2)));
#endif
}
// hardware transcendentals for the fused encoder-side kernels (v_exp_f32 / v_rsq_f32: ~1 ulp); the one-kernel-per-op plan and
// the oracle keep the libm forms.  The simulator mirrors the formulas with libm calls.
__device__ __forceinline__ float exp_fast_f32(float x) {
#ifdef ESMI_WAVESIM
    return exp2f(x * 1.4426950408889634f);
#else
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
#endif
}
__device__ __forceinline__ float rsqrt_fast_f32(float x) {
#ifdef ESMI_WAVESIM
    return 1.0f / sqrtf(x);
#else
    return __builtin_amdgcn_rsqf(x);
#endif
}
__device__ __forceinline__ float rcp_fast_f32(float x) {
#ifdef ESMI_WAVESIM
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}
// exact-erf GELU to 1.5e-7: erf by Abramowitz & Stegun 7.1.26 (|err| <= 1.5e-7), branch-free: 1 rcp + 1 exp + 7 fma
__device__ __forceinline__ float gelu_fast_f32(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = rcp_fast_f32(fmaf(0.3275911f, z, 1.0f));
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float erf_abs = 1.0f - pl * t * exp_fast_f32(-z * z);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}
__device__ __forceinline__ float ident_f32(float x) { return x; }
__device__ __forceinline__ float gelu_erf_f32(float x) {  // nn.GELU() default = exact erf form
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_TANH = 3 };
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.0f);
        case ACT_GELU: return gelu_erf_f32(v);
        case ACT_TANH: return tanh_f32(v);
        default: return v;
    }
}

// ---- LayerNorm over the C = 32*NT columns of each tile row, in the MFMA C/D register layout.
// v[nt][r] holds (row = tile_row(r), col = 32*nt + (lane&31)).  Two-pass (mean, then centred
// variance), biased variance, eps inside the sqrt -- nn.LayerNorm semantics.
template <int NT>
__device__ __forceinline__ void layernorm_tile(f32x16 (&v)[NT], const float* __restrict__ g,
                                               const float* __restrict__ b, int lane, float eps = 1e-5f) {
    const float inv_c = 1.0f / (float)(32 * NT);
    float gg[NT], bb[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        gg[nt] = g[32 * nt + (lane & 31)];
        bb[nt] = b[32 * nt + (lane & 31)];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float s = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) s += v[nt][r];
        const float mean = row_sum32(s) * inv_c;
        float q = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float d = v[nt][r] - mean;
            q = fmaf(d, d, q);
        }
        const float var = row_sum32(q) * inv_c;
        const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) v[nt][r] = fmaf((v[nt][r] - mean) * rstd, gg[nt], bb[nt]);
    }
}

// same, gain / bias already in registers (gg[nt], bb[nt] = values of column 32*nt + (lane&31))
template <int NT>
__device__ __forceinline__ void layernorm_tile_regs(f32x16 (&v)[NT], const float (&gg)[NT], const float (&bb)[NT],
                                                    float eps = 1e-5f) {
    const float inv_c = 1.0f / (float)(32 * NT);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float s = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) s += v[nt][r];
        const float mean = row_sum32(s) * inv_c;
        float q = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float d = v[nt][r] - mean;
            q = fmaf(d, d, q);
        }
        const float var = row_sum32(q) * inv_c;
        const float rstd = rsqrt_fast_f32(var + eps);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) v[nt][r] = fmaf((v[nt][r] - mean) * rstd, gg[nt], bb[nt]);
    }
}

// An offset the optimiser must treat as freshly computed here.  Used on the lane-dependent part of LDS addresses
// inside loops: without it LLVM's LICM hoists every `base + constant` address of the loop body into its own
// loop-invariant VGPR (measured: ~100 address registers in the mel decoder, 80+ of them spilled at a 128-VGPR
// budget), and instruction selection can then no longer fold the constants into the 16-bit DS offset field.
__device__ __forceinline__ int opaque_i(int v) {
#ifndef ESMI_WAVESIM
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// ---- bounds-checked buffer access (raw buffer resources, stride 0).  An access whose byte offset is not inside
// [0, bytes) reads 0 / is dropped BY THE HARDWARE, so ragged tile edges need no branch: a branch around a load or a
// store makes hipcc's s_waitcnt counting fall back to vmcnt(0) for everything still in flight (the number of younger
// operations is no longer known), which serialises every software pipeline around it.  `kBufOOB` is an offset that is
// always out of range (tensors here are < 2 GiB).  The base must be wave-uniform.
constexpr unsigned kBufOOB = 0x80000000u;
#ifdef ESMI_WAVESIM
struct BufRsrc { const char* base; unsigned bytes; };
__device__ __forceinline__ BufRsrc make_rsrc(const void* p, long bytes) {
    BufRsrc r = {static_cast<const char*>(p), p ? (unsigned)(bytes < 0 ? 0 : (bytes > 0x7fffffffL ? 0x7fffffffL : bytes)) : 0u};
    return r;
}
__device__ __forceinline__ float buf_ld(const BufRsrc& r, unsigned off) {
    return (off < r.bytes && off + 4u <= r.bytes) ? *reinterpret_cast<const float*>(r.base + off) : 0.0f;
}
__device__ __forceinline__ f32x4 buf_ld4(const BufRsrc& r, unsigned off) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return (off < r.bytes && off + 16u <= r.bytes) ? *reinterpret_cast<const f32x4*>(r.base + off) : z;
}
__device__ __forceinline__ unsigned buf_ld_u8(const BufRsrc& r, unsigned off) {
    return off < r.bytes ? (unsigned)*reinterpret_cast<const unsigned char*>(r.base + off) : 0u;
}
// 16 bytes at byte offset voff (per lane) + soff (wave-uniform): the uniform part travels in an SGPR, so one lane-offset VGPR
// serves every access of a kernel to the same buffer (no 64-bit per-lane pointers kept alive across loops)
__device__ __forceinline__ f32x4 buf_ld4s(const BufRsrc& r, unsigned voff, unsigned soff) { return buf_ld4(r, voff + soff); }
__device__ __forceinline__ void buf_st(const BufRsrc& r, unsigned off, float v) {
    if (off < r.bytes && off + 4u <= r.bytes) *reinterpret_cast<float*>(const_cast<char*>(r.base) + off) = v;
}
__device__ __forceinline__ void buf_st_i(const BufRsrc& r, unsigned off, int v) {
    if (off < r.bytes && off + 4u <= r.bytes) *reinterpret_cast<int*>(const_cast<char*>(r.base) + off) = v;
}
__device__ __forceinline__ void lds_wave_sync() { wavesim::shfl_i(0, 0); }   // a wave-level collective: all 64 fibers arrive
#else
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc make_rsrc(const void* p, long bytes) {
    const int n = p ? (int)(bytes < 0 ? 0 : (bytes > 0x7fffffffL ? 0x7fffffffL : bytes)) : 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, n, 0x00020000);   // gfx9 raw buffer, dword 3
}
__device__ __forceinline__ float buf_ld(BufRsrc r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ f32x4 buf_ld4(BufRsrc r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}
__device__ __forceinline__ unsigned buf_ld_u8(BufRsrc r, unsigned off) {
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b8(r, (int)off, 0, 0);
}
// 16 bytes at byte offset voff (per lane) + soff (wave-uniform): the uniform part travels in an SGPR, so one lane-offset VGPR
// serves every access of a kernel to the same buffer (no 64-bit per-lane pointers kept alive across loops)
__device__ __forceinline__ f32x4 buf_ld4s(BufRsrc r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, __builtin_amdgcn_readfirstlane((int)soff), 0));
}
#ifndef ESMI_ST_AUX
#define ESMI_ST_AUX 0   // cache policy bits of the tensor-output stores (gfx94x: 1 = sc0, 2 = nt, 16 = sc1)
#endif
__device__ __forceinline__ void buf_st(BufRsrc r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, ESMI_ST_AUX);
}
__device__ __forceinline__ void buf_st_i(BufRsrc r, unsigned off, int v) {
    __builtin_amdgcn_raw_buffer_store_b32((unsigned)v, r, (int)off, 0, 0);
}
// Hand-off through LDS between lanes of ONE wave (tile_store -> A-fragment reads): LDS operations of a wave execute
// in issue order, so only the compiler has to be kept from reordering; no s_barrier, and global loads in flight
// (weight prefetches) stay in flight.
__device__ __forceinline__ void lds_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#endif

// Pin the instruction order at this point.  hipcc's scheduler sinks prefetch loads down to their first use to shorten
// live ranges (seen in the chain kernels: `global_load; s_waitcnt vmcnt(0); 4 x v_mfma` per k-step, i.e. every
// software pipeline collapsed); a scheduling barrier after each prefetch block keeps the loads where they were written.
__device__ __forceinline__ void sched_fence() {
#ifndef ESMI_WAVESIM
    __builtin_amdgcn_sched_barrier(0);
#endif
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

}  // namespace esmi
