// Vector types, the software binary16 conversions and the constants of the split-f16 contractions: pure C++ shared by the target
// shim (wavesim_shim.h, which includes this file behind the target's runtime header), the device primitives (esmi_dev.h) and the
// weight packers.  No target-specific code here.
#pragma once

namespace esmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct bf16x3 { u32x4 hi, mid, lo; };
struct f16x2p { u32x4 h1, h2; };
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

}  // namespace esmi
