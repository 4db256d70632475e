// THE target shim: every line of the kernel sources that differs between the product build (hipcc --offload-arch=gfx950: CDNA4
// instructions) and the test-only CPU wave simulator (tools/wavesim, host clang++ -DESMI_WAVESIM: lane-accurate emulation, so that
// index math / MFMA layouts / LDS protocols can be checked without a GPU) lives in this file.  The kernel headers are written once
// against these primitives and contain no ESMI_WAVESIM branch.  The simulator is a test tool bound only from tests/simlib.py; the
// product package loads libesmi.so (real HIP) or raises.
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

#include "esmi_types.h"

#ifndef ESMI_CHAIN_SPLIT
#define ESMI_CHAIN_SPLIT 1   // weight GEMMs of the encoder-side chain kernels: 1 = split-f16x2 (3 f16 MFMAs per 16 channels), 0 = fp32 MFMA
#endif

namespace esmi {

// ---- what the two targets configure differently (the simulator runs a fiber per thread: small workgroups / few of them, and the
// LDS-staged kernels at every size so that the tests reach them)
#ifdef ESMI_WAVESIM
constexpr const char* kBackendName = "wavesim";
constexpr int kReduceGroups = 4;            // train_ops.h: waves per deterministic-reduction workgroup
constexpr int kLossBlocks = 8;              // train_ops.h: workgroups of the loss kernel
constexpr int kAttnLdsMinHeadsDefault = 1;  // tu_attention.hip: (utterance, head) pairs from which attention is LDS-staged
constexpr int kGemmLdsMinRowsDefault = 1;   // tu_convgemm.hip: rows from which a GEMM is LDS-staged
constexpr int kPwGemmMinRows = 1;           // tu_convgemm.hip: rows from which a k = 1 GEMM keeps its whole weight in LDS (pwgemm.h)
#else
constexpr const char* kBackendName = "hip:gfx950";
constexpr int kReduceGroups = 16;
constexpr int kLossBlocks = 256;
constexpr int kAttnLdsMinHeadsDefault = 128;   // enough (utterance, head) workgroups to occupy the chip at one per CU
constexpr int kGemmLdsMinRowsDefault = 2048;
constexpr int kPwGemmMinRows = 32768;       // (one 8-wave workgroup per CU, >= 4 row tiles per wave... below that the staging is not amortised)
#endif

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
// measured on K = 128 dot products, max error 3.0e-6 vs 8.5e-6 for sequential fp32 accumulation (HISTORY.md 3.1).
//   A[i = lane&31][k = 8*(lane>>5) + (0..7)],  B[k = 8*(lane>>5) + (0..7)][j = lane&31]   (8 bf16 = 4 dwords per lane)
//   D as for the 32x32x2 fp32 MFMA
__device__ __forceinline__ unsigned pack_hi16(unsigned even, unsigned odd) {   // {odd[31:16], even[31:16]}
#ifdef ESMI_WAVESIM
    return (odd & 0xFFFF0000u) | (even >> 16);
#else
    return __builtin_amdgcn_perm(odd, even, 0x07060302);
#endif
}
__device__ __forceinline__ f32x16 mfma32_bf16(const u32x4& a, const u32x4& b, f32x16 c) {
#ifdef ESMI_WAVESIM
    return wavesim::mfma_32x32x16_bf16(a, b, c);
#else
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#endif
}
// ---- the same idea on the f16 matrix pipe with HALF the products: x = h1 + h2 (two binary16 pieces, 11 + 11 significand
// bits; h1 by round-toward-zero so the residual x - h1 is exact), weights pre-scaled by 2^8 and pre-split the same way
// (round to nearest), a.w ~= h1.w1 + h1.w2 + h2.w1: the dropped h2.w2 term and the pieces' rounding are ~2^-22 relative,
// below the fp32 accumulation error of a K >= 32 contraction (numpy emulation, K = 128: max error 0.8-3.2e-6 vs 1.4-4.6e-6
// for sequential fp32 accumulation, HISTORY.md 3.1).  Operands must be inside the binary16 range: |a| < 65504 and
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
// a * b for a, b < 2^24 (v_mul_u32_u24: full rate; v_mul_lo_u32 is quarter rate)
__device__ __forceinline__ unsigned mul24u(unsigned a, unsigned b) {
#ifdef ESMI_WAVESIM
    return a * b;
#else
    return __umul24(a, b);
#endif
}
// lane `src`'s value for the whole wave, `src` wave-uniform (v_readlane_b32; every lane of the wave must arrive)
__device__ __forceinline__ int bcast_i(int v, int src) {
#ifdef ESMI_WAVESIM
    return wavesim::shfl_i(v, src);
#else
    return __builtin_amdgcn_readlane(v, src);
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

// lane l <- lane l ^ 16 (VALU only: v_permlane16_swap, gfx950).  The select form: ROCm 7.2 folds `swap(x, x); a + b` of the builtin's
// two results into x + x (profiles/r04_probes/chain_layernorm_round4.md); selecting one result per lane compiles correctly.
__device__ __forceinline__ float swap16_f(float v) {
#ifdef ESMI_WAVESIM
    return wavesim::shfl(v, lane_id_raw() ^ 16);
#else
    const unsigned x = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);   // r[0]: odd rows <- the even rows below them; r[1]: even rows <- the odd rows above
    return __builtin_bit_cast(float, (lane_id_raw() & 16) ? r[0] : r[1]);
#endif
}
// rows one down / one up inside the 16 lanes of a DPP row: lane (i, g) <- lane (i -+ 1, g); the lane at the row's edge takes `edge`
// (v_mov_b32_dpp row_shr:1 / row_shl:1 with bound_ctrl off: an out-of-row source leaves `old` in place)
__device__ __forceinline__ unsigned row_dn_u(unsigned v, unsigned edge) {
#ifdef ESMI_WAVESIM
    const int l = lane_id_raw();
    const unsigned s = (unsigned)wavesim::shfl_i((int)v, (l & 15) ? l - 1 : l);
    return (l & 15) ? s : edge;
#else
    return (unsigned)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x111, 0xF, 0xF, false);
#endif
}
__device__ __forceinline__ unsigned row_up_u(unsigned v, unsigned edge) {
#ifdef ESMI_WAVESIM
    const int l = lane_id_raw();
    const unsigned s = (unsigned)wavesim::shfl_i((int)v, (l & 15) != 15 ? l + 1 : l);
    return (l & 15) != 15 ? s : edge;
#else
    return (unsigned)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x101, 0xF, 0xF, false);
#endif
}
// a device-side global pointer a translation unit owns (the decoder's clock probe): a plain global in the simulator
#ifdef ESMI_WAVESIM
#define ESMI_DEVICE_GLOBAL_PTR(type, name) static type* name = nullptr
#define ESMI_STORE_DEVICE_GLOBAL_PTR(name, value) ((name) = (value), 0)
#else
#define ESMI_DEVICE_GLOBAL_PTR(type, name) static __device__ type* name = nullptr
#define ESMI_STORE_DEVICE_GLOBAL_PTR(name, value) ((int)hipMemcpyToSymbol(HIP_SYMBOL(name), &(value), sizeof(value)))
#endif

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
// the two clocks a wave can read: the shader clock (s_memtime: advances with the frequency the CU runs at) and the constant
// 100 MHz clock (s_memrealtime); the simulator has neither
__device__ __forceinline__ long long clock_shader() {
#ifdef ESMI_WAVESIM
    return 0;
#else
    return (long long)__builtin_amdgcn_s_memtime();
#endif
}
__device__ __forceinline__ long long clock_real100() {
#ifdef ESMI_WAVESIM
    return 0;
#else
    return (long long)__builtin_amdgcn_s_memrealtime();
#endif
}
// a value the optimiser must treat as freshly computed (see opaque_i in esmi_dev.h)
__device__ __forceinline__ int opaque_i(int v) {
#ifndef ESMI_WAVESIM
    asm volatile("" : "+v"(v));
#endif
    return v;
}
// a wave-uniform value made a scalar register
__device__ __forceinline__ int uniform_i(int v) {
#ifdef ESMI_WAVESIM
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}
// LDS-DMA (global_load_lds_dwordx4): 16 bytes per lane from this lane's global address straight into LDS at `lds_wave_base` + 16 lane
// (the LDS address is wave-uniform + the lane's slot: that is how the instruction addresses), no staging registers
__device__ __forceinline__ void lds_dma16(const void* gsrc_lane, void* lds_wave_base, int lane) {
#ifdef ESMI_WAVESIM
    reinterpret_cast<f32x4*>(lds_wave_base)[lane] = *reinterpret_cast<const f32x4*>(gsrc_lane);
#else
    (void)lane;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
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
// (4 bytes at per-lane offset voff + wave-uniform soff, as buf_ld4s)
__device__ __forceinline__ void buf_st_s(const BufRsrc& r, unsigned voff, unsigned soff, float v) { buf_st(r, voff == kBufOOB ? kBufOOB : voff + soff, v); }
__device__ __forceinline__ void buf_st4(const BufRsrc& r, unsigned off, const f32x4& v) {
    if (off < r.bytes && off + 16u <= r.bytes) *reinterpret_cast<f32x4*>(const_cast<char*>(r.base) + off) = v;
}
__device__ __forceinline__ void lds_wave_sync() { wavesim::shfl_i(0, 0); }   // a wave-level collective: all 64 fibers arrive
__device__ __forceinline__ void wait_vm0() {}                                // (LDS-DMA is synchronous in the simulator)
__device__ __forceinline__ void wg_sync_lds() { __syncthreads(); }
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
__device__ __forceinline__ void buf_st(BufRsrc r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 0);   // (non-temporal / sc1 stores of the outputs measured no gain: profiles/r03_d)
}
__device__ __forceinline__ void buf_st_i(BufRsrc r, unsigned off, int v) {
    __builtin_amdgcn_raw_buffer_store_b32((unsigned)v, r, (int)off, 0, 0);
}
// (4 bytes at per-lane offset voff + wave-uniform soff, as buf_ld4s; voff = kBufOOB stays out of range for every soff < 2 GiB)
__device__ __forceinline__ void buf_st_s(BufRsrc r, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, __builtin_amdgcn_readfirstlane((int)soff), 0);
}
__device__ __forceinline__ void buf_st4(BufRsrc r, unsigned off, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)off, 0, 0);
}
// every global / LDS-DMA operation of this wave has completed (the LDS-DMA writes are visible to the wave; a workgroup barrier
// behind it makes them visible to the others)
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Workgroup barrier for LDS hand-offs only: this wave's LDS operations are complete, then s_barrier.  Unlike __syncthreads() it does
// NOT drain the vector-memory queue, so LDS-DMA weight copies for a later stage (and output stores) stay in flight across it; a
// barrier that publishes DMA'd data is `wait_vm0(); wg_sync_lds();`.
__device__ __forceinline__ void wg_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
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


// ---- host side
// current device ordinal, or -(error code)
inline int current_device() {
#ifdef ESMI_WAVESIM
    return 0;
#else
    int dev = 0;
    const hipError_t e = hipGetDevice(&dev);
    return e == hipSuccess ? dev : -(int)e;
#endif
}
inline int set_max_dynamic_lds(const void* fn, int bytes) {   // 0 or an error code
#ifdef ESMI_WAVESIM
    (void)fn; (void)bytes;
    return 0;
#else
    return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
#endif
}
// read one int32 the kernels of `stream` wrote at `dev_flag` (drains the stream); 0 or an error code
inline int read_device_flag(const int* dev_flag, hipStream_t stream, int* out) {
#ifdef ESMI_WAVESIM
    (void)stream;
    *out = *dev_flag;
    return 0;
#else
    hipError_t e = hipStreamSynchronize(stream);
    if (e == hipSuccess) e = hipMemcpy(out, dev_flag, sizeof(int), hipMemcpyDeviceToHost);
    return (int)e;
#endif
}
#if ESMI_RANGE_CHECK
// point this translation unit's copy of the device-side range-flag pointer at `flag` (static: the pointer it writes is this unit's own,
// so an out-of-line copy must not be shared between units)
static inline int store_range_flag_pointer(int* flag) {
#ifdef ESMI_WAVESIM
    g_esmi_range_flag = flag;
    return 0;
#else
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_esmi_range_flag), &flag, sizeof(flag));
#endif
}
#endif

}  // namespace esmi
