// Device-side primitives shared by the esmi kernels (gfx950 / CDNA4, wave64).
//
// The kernels are written once, against the handful of primitives below.  The product build
// (hipcc --offload-arch=gfx950) maps them to the CDNA4 instructions; the test-only wave
// simulator (tools/wavesim, -DESMI_WAVESIM, host clang++) maps them to lane-accurate CPU
// emulation so that index math / MFMA layouts / LDS protocols can be checked without a GPU.
#pragma once
#include "wavesim_shim.h"

namespace esmi {

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

// row of accumulator register r inside a 32-row MFMA tile
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

// ---- activations (fp32; |err| <= ~2e-7 absolute, far inside the 1e-4 parity budget): tanh_f32 and the fast forms are in wavesim_shim.h
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


// (Round 4 used a form of this with the reductions of the 16 rows done stage by stage in the two encoder-block kernels: ~1 % faster,
// but the same change gave wrong rows on the GPU inside enc_fuse_va_kernel for a reason that was never found
// (profiles/r04_probes/chain_layernorm_round4.md).  Round 5 replaced those kernels on their hot shapes by the chain16 kernels -- whose
// LayerNorm needs no per-row reductions at all -- and the kernels that remain as fallbacks use the row-by-row form above everywhere.)

// An offset the optimiser must treat as freshly computed here (`opaque_i`, wavesim_shim.h).  Used on the lane-dependent part of LDS
// addresses inside loops: without it LLVM's LICM hoists every `base + constant` address of the loop body into its own
// loop-invariant VGPR (measured: ~100 address registers in the mel decoder, 80+ of them spilled at a 128-VGPR budget), and
// instruction selection can then no longer fold the constants into the 16-bit DS offset field.

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

}  // namespace esmi
