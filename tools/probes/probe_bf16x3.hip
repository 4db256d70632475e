// Development probe for round 2: cost of an fp32-accurate contraction step on the bf16 matrix pipe.
// One "set" = K = 16 of a 32x32 output tile: fp32 path = 8 x v_mfma_f32_32x32x2_f32; split path = truncate-split the A
// fragment (8 fp32 per lane from LDS) into hi/mid/lo bf16 on the fly + 6 x v_mfma_f32_32x32x16_bf16 (B pre-split).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, u32x4& hi, u32x4& mid, u32x4& lo) {
    float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned xb = __builtin_bit_cast(unsigned, x[j]);
        h[j] = xb & 0xFFFF0000u;
        const float r1 = x[j] - __builtin_bit_cast(float, h[j]);
        m[j] = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
        const float r2 = r1 - __builtin_bit_cast(float, m[j]);
        l[j] = __builtin_bit_cast(unsigned, r2);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // two bf16 (upper halves) per dword
        hi[j] = __builtin_amdgcn_perm(h[2 * j + 1], h[2 * j], 0x07060302);
        mid[j] = __builtin_amdgcn_perm(m[2 * j + 1], m[2 * j], 0x07060302);
        lo[j] = __builtin_amdgcn_perm(l[2 * j + 1], l[2 * j], 0x07060302);
    }
}
template <int MODE>   // 0: fp32 MFMA, 1: split-bf16 with on-the-fly A split, 2: split-bf16 MFMAs only (A pre-split)
__global__ __launch_bounds__(512, 1) void probe(float* out, long long* cyc, int sets) {
    extern __shared__ float lds[];
    for (int e = threadIdx.x; e < 132 * 132; e += 512) lds[e] = 1.0f + 1e-3f * (e % 97);
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 31, kb = lane >> 5;
    f32x16 acc[2][2] = {{{0}, {0}}, {{0}, {0}}};   // 2 row tiles x (main, cross) accumulators
    u32x4 bh = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, bm = bh, bl = bh;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < sets; ++s) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float* row = lds + (2 + 32 * mt + i) * 132 + ((16 * s) & 127) + 8 * kb;
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(row), x1 = *reinterpret_cast<const f32x4*>(row + 4);
            if (MODE == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[q], 1.0f, acc[mt][0], 0, 0, 0);
                    acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[q], 1.0f, acc[mt][1], 0, 0, 0);
                }
            } else {
                u32x4 ah, am, al;
                if (MODE == 1) split8(x0, x1, ah, am, al);
                else { ah = __builtin_bit_cast(u32x4, x0); am = __builtin_bit_cast(u32x4, x1); al = ah; }
                const bf16x8 A0 = __builtin_bit_cast(bf16x8, ah), A1 = __builtin_bit_cast(bf16x8, am), A2 = __builtin_bit_cast(bf16x8, al);
                const bf16x8 B0 = __builtin_bit_cast(bf16x8, bh), B1 = __builtin_bit_cast(bf16x8, bm), B2 = __builtin_bit_cast(bf16x8, bl);
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B1, acc[mt][1], 0, 0, 0);
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B0, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc[mt][1], 0, 0, 0);
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B2, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B0, acc[mt][1], 0, 0, 0);
            }
        }
    }
    float r = 0;
    for (int q = 0; q < 16; ++q) r += acc[0][0][q] + acc[0][1][q] + acc[1][0][q] + acc[1][1][q];
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE> void run(const char* name, int sets) {
    const int n = 256;
    float* out; long long* cyc;
    hipMalloc(&out, n * 512 * 4); hipMalloc(&cyc, n * 8 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 132 * 4);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE>), dim3(n), dim3(512), 132 * 132 * 4, 0, out, cyc, sets);
    hipDeviceSynchronize();
    std::vector<long long> c(n * 8);
    hipMemcpy(c.data(), cyc, n * 8 * 8, hipMemcpyDeviceToHost);
    double a = 0; for (auto v : c) a += v;
    printf("%-46s %7.1f cycles per (K=16, 2 row tiles) set per wave, 2 waves per SIMD\n", name, a / (n * 8) / sets);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("fp32: 16 x v_mfma_f32_32x32x2_f32", 400);
    run<2>("bf16 pipe only: 12 x v_mfma_f32_32x32x16_bf16", 400);
    run<1>("split-bf16: A split on the fly + 12 MFMAs", 400);
    return 0;
}
