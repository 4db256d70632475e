// Development probe: does a stream of v_mfma_f32_32x32x2_f32 from one wave slow down VALU / LDS / transcendental work of
// ANOTHER wave on the same SIMD?  One workgroup of 8 waves per CU: waves 0-3 (one per SIMD) run role A, waves 4-7 role B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { IDLE = 0, MFMA = 1, VALU = 2, LDS = 3, TRANS = 4, MIX_V = 5, MIX_L = 6, MIX_T = 7 };
template <int ROLE>
__device__ __forceinline__ float work(int iters, float* lds, int prio) {
    if (prio == 1) __builtin_amdgcn_s_setprio(1);
    if (prio == 2) __builtin_amdgcn_s_setprio(2);
    float x = threadIdx.x * 1e-3f, y = 1.0001f, s = 0.f;
    if (ROLE == MFMA) {
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    } else if (ROLE == VALU) {
        float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
        for (int i = 0; i < iters * 8; ++i) {   // 8 independent fma chains, 64 fma per trip
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v0 = fmaf(v0, y, x); v1 = fmaf(v1, y, x); v2 = fmaf(v2, y, x); v3 = fmaf(v3, y, x);
                v4 = fmaf(v4, y, x); v5 = fmaf(v5, y, x); v6 = fmaf(v6, y, x); v7 = fmaf(v7, y, x);
            }
        }
        s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    } else if (ROLE == LDS) {
        const f32x4* p = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63) * 33;
        f32x4 acc = {0, 0, 0, 0};
        for (int i = 0; i < iters * 4; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += p[(i * 8 + u) & 31];
        }
        s = acc[0] + acc[1] + acc[2] + acc[3];
    } else if (ROLE == TRANS) {
        float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3;
        for (int i = 0; i < iters * 4; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v0 = __builtin_amdgcn_exp2f(v0) * 0.5f; v1 = __builtin_amdgcn_exp2f(v1) * 0.5f;
                v2 = __builtin_amdgcn_rcpf(v2 + 2.f); v3 = __builtin_amdgcn_rcpf(v3 + 2.f);
            }
        }
        s = v0 + v1 + v2 + v3;
    }
    if (ROLE == MIX_V || ROLE == MIX_L || ROLE == MIX_T) {   // same wave: 4 MFMAs, then independent filler work, per trip
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3;
        const f32x4* p = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63) * 33;
        f32x4 acc = {0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
            if (ROLE == MIX_V) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { v0 = fmaf(v0, y, x); v1 = fmaf(v1, y, x); v2 = fmaf(v2, y, x); v3 = fmaf(v3, y, x); }
            } else if (ROLE == MIX_L) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc += p[(i * 4 + u) & 31];
            } else {
                v0 = __builtin_amdgcn_exp2f(v0) * 0.5f; v1 = __builtin_amdgcn_exp2f(v1) * 0.5f;
                v2 = __builtin_amdgcn_rcpf(v2 + 2.f); v3 = __builtin_amdgcn_rcpf(v3 + 2.f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
        s += v0 + v1 + v2 + v3 + acc[0] + acc[1] + acc[2] + acc[3];
    }
    return s;
}
template <int RA, int RB>
__global__ __launch_bounds__(512, 1) void probe(float* out, long long* cyc, int iters, int prio_b) {
    extern __shared__ float lds[];
    for (int e = threadIdx.x; e < 64 * 33 * 4; e += 512) lds[e] = 1.0f;
    __syncthreads();
    const int w = threadIdx.x >> 6;
    long long t0 = __builtin_amdgcn_s_memtime();
    float s = (w < 4) ? work<RA>(iters, lds, 0) : work<RB>(iters, lds, prio_b);
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}
template <int RA, int RB> void run(const char* name, int iters, int prio_b) {
    const int n = 256;
    float* out; long long* cyc;
    hipMalloc(&out, n * 512 * 4); hipMalloc(&cyc, n * 8 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<RA, RB>), dim3(n), dim3(512), 64 * 33 * 16, 0, out, cyc, iters, prio_b);
    hipDeviceSynchronize();
    std::vector<long long> c(n * 8);
    hipMemcpy(c.data(), cyc, n * 8 * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < n; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += c[i * 8 + w];
    printf("%-34s prio_b=%d: waves 0-3 %8.0f cycles, waves 4-7 %8.0f cycles\n", name, prio_b, a / (n * 4), b / (n * 4));
    hipFree(out); hipFree(cyc);
}
int main() {
    const int it = 200;
    run<MFMA, IDLE>("A=mfma alone", it, 0);
    run<IDLE, VALU>("B=valu alone", it, 0);
    run<IDLE, LDS>("B=lds alone", it, 0);
    run<IDLE, TRANS>("B=trans alone", it, 0);
    run<MFMA, MFMA>("A=mfma B=mfma", it, 0);
    run<MIX_V, IDLE>("A=4 mfma + 16 fma per trip", it, 0);
    run<MIX_L, IDLE>("A=4 mfma + 4 ds_read_b128", it, 0);
    run<MIX_T, IDLE>("A=4 mfma + 2 exp + 2 rcp", it, 0);
    run<MIX_V, MIX_V>("A=B=4 mfma + 16 fma", it, 0);
    run<MIX_L, MIX_L>("A=B=4 mfma + 4 ds_read", it, 0);
    run<MIX_T, MIX_T>("A=B=4 mfma + 2exp+2rcp", it, 0);
    for (int p = 0; p < 1; ++p) {
        run<MFMA, VALU>("A=mfma B=valu", it, p);
        run<MFMA, LDS>("A=mfma B=lds", it, p);
        run<MFMA, TRANS>("A=mfma B=trans", it, p);
    }
    return 0;
}
