// wavesim -- a lane-accurate CPU simulator for the subset of HIP the esmi kernels use.
//
// DEVELOPMENT / TEST TOOL ONLY.  It exists because the build container has no GPU and GPU
// minutes are scarce: the unmodified kernel sources (efficientspeech_amd/csrc/*.hip) are compiled
// by the host clang++ with -DESMI_WAVESIM into libesmi_sim.so, where every GPU thread is a
// cooperative fiber, a wavefront is 64 fibers exchanging operands at each collective
// (MFMA, shuffle), and a workgroup is the set of fibers sharing __shared__ storage and
// __syncthreads().  Tests under tests/ run the kernels' index/MFMA-layout/LDS logic against the
// oracle on the CPU.  The product package (efficientspeech_amd) never loads this library: it
// binds only libesmi.so (real HIP, gfx950) and raises if that is missing.
//
// MFMA lane layouts follow /opt/skills/guides/cdna_hip_programming.md §3:
//   16x16x4f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D: col=l&15, row=4*(l>>4)+reg
//   32x32x2f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D: col=l&31, row=(reg&3)+8*(reg>>2)+4*(l>>5)
// and the numerics are the documented k-ordered fp32 fmaf chain.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
enum { hipMemcpyDeviceToDevice = 3 };
template <class F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }

namespace wavesim {

struct Fiber;
Fiber* cur();
uint3_ tid_of(const Fiber*);
uint3_ bid();
dim3 bdim();
dim3 gdim();
void* dyn_lds();
void sync_block();
// deposit n 32-bit words for this lane; returns pointer to the 64-lane table [lane][n] once all
// 64 lanes of the wave have deposited.  The table stays valid until this lane's next collective.
const uint32_t* wave_exchange(const uint32_t* words, int n);
int lane_id();
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body);

}  // namespace wavesim

#define threadIdx (wavesim::tid_of(wavesim::cur()))
#define blockIdx (wavesim::bid())
#define blockDim (wavesim::bdim())
#define gridDim (wavesim::gdim())

static inline void __syncthreads() { wavesim::sync_block(); }

typedef float __attribute__((ext_vector_type(4))) ws_f32x4;
typedef float __attribute__((ext_vector_type(16))) ws_f32x16;

namespace wavesim {

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline ws_f32x16 mfma_32x32x2(float a, float b, ws_f32x16 c) {
    uint32_t w[2] = {f2u(a), f2u(b)};
    const uint32_t* t = wave_exchange(w, 2);
    const int l = lane_id();
    ws_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(u2f(t[(row + 32 * k) * 2 + 0]), u2f(t[(col + 32 * k) * 2 + 1]), acc);
        d[r] = acc;
    }
    return d;
}

// v_mfma_f32_32x32x16_bf16: lane (i = l&31, kb = l>>5) holds A[i][8kb..8kb+7] / B[8kb..8kb+7][i] as 8 bf16 in 4 dwords
template <class V4>
static inline ws_f32x16 mfma_32x32x16_bf16(const V4& a, const V4& b, ws_f32x16 c) {
    uint32_t w[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    const uint32_t* t = wave_exchange(w, 8);
    const int l = lane_id();
    auto elem = [&](int lane, int base, int k) {   // bf16 element k (0..7) of a lane's operand -> float
        const uint32_t d = t[lane * 8 + base + (k >> 1)];
        return u2f((k & 1) ? (d & 0xFFFF0000u) : (d << 16));
    };
    ws_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) acc = fmaf(elem(row + 32 * (k >> 3), 0, k & 7), elem(col + 32 * (k >> 3), 4, k & 7), acc);
        d[r] = acc;
    }
    return d;
}

// v_mfma_f32_32x32x16_f16: the same operand layout with binary16 elements
static inline float ws_half_to_float(uint32_t h) {
    const uint32_t sign = (h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3FFu;
    if (e == 31u) return u2f(sign | 0x7F800000u | (m << 13));
    if (e == 0u) { const float v = (float)m * 5.9604644775390625e-08f; return (h & 0x8000u) ? -v : v; }
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}
template <class V4>
static inline ws_f32x16 mfma_32x32x16_f16(const V4& a, const V4& b, ws_f32x16 c) {
    uint32_t w[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    const uint32_t* t = wave_exchange(w, 8);
    const int l = lane_id();
    auto elem = [&](int lane, int base, int k) {
        const uint32_t d = t[lane * 8 + base + (k >> 1)];
        return ws_half_to_float((k & 1) ? (d >> 16) : (d & 0xFFFFu));
    };
    ws_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) acc = fmaf(elem(row + 32 * (k >> 3), 0, k & 7), elem(col + 32 * (k >> 3), 4, k & 7), acc);
        d[r] = acc;
    }
    return d;
}

// v_mfma_f32_16x16x32_f16: lane (i = l&15, kb = l>>4) holds A[i][8kb..8kb+7] / B[8kb..8kb+7][i]; D as the 16x16x4 fp32 MFMA
template <class V4>
static inline ws_f32x4 mfma_16x16x32_f16(const V4& a, const V4& b, ws_f32x4 c) {
    uint32_t w[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    const uint32_t* t = wave_exchange(w, 8);
    const int l = lane_id();
    auto elem = [&](int lane, int base, int k) {
        const uint32_t d = t[lane * 8 + base + (k >> 1)];
        return ws_half_to_float((k & 1) ? (d >> 16) : (d & 0xFFFFu));
    };
    ws_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) acc = fmaf(elem(row + 16 * (k >> 3), 0, k & 7), elem(col + 16 * (k >> 3), 4, k & 7), acc);
        d[r] = acc;
    }
    return d;
}

static inline ws_f32x4 mfma_16x16x4(float a, float b, ws_f32x4 c) {
    uint32_t w[2] = {f2u(a), f2u(b)};
    const uint32_t* t = wave_exchange(w, 2);
    const int l = lane_id();
    ws_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(u2f(t[(row + 16 * k) * 2 + 0]), u2f(t[(col + 16 * k) * 2 + 1]), acc);
        d[r] = acc;
    }
    return d;
}

static inline float shfl_xor(float v, int mask) {
    uint32_t w = f2u(v);
    const uint32_t* t = wave_exchange(&w, 1);
    return u2f(t[lane_id() ^ mask]);
}
static inline float shfl(float v, int src) {
    uint32_t w = f2u(v);
    const uint32_t* t = wave_exchange(&w, 1);
    return u2f(t[src & 63]);
}
static inline int shfl_i(int v, int src) {
    uint32_t w = (uint32_t)v;
    const uint32_t* t = wave_exchange(&w, 1);
    return (int)t[src & 63];
}
static inline unsigned long long ballot(bool pred) {
    uint32_t w = pred ? 1u : 0u;
    const uint32_t* t = wave_exchange(&w, 1);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) m |= (unsigned long long)(t[l] & 1u) << l;
    return m;
}
static inline int shfl_up_i(int v, int delta) {   // lanes < delta keep their own value
    uint32_t w = (uint32_t)v;
    const uint32_t* t = wave_exchange(&w, 1);
    const int l = lane_id();
    return l >= delta ? (int)t[l - delta] : v;
}

}  // namespace wavesim

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline int atomicMax(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
