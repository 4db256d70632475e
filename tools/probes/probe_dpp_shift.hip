// Development probe: cross-lane shifts a frames-along-lanes depthwise conv could use on gfx950 -- semantics and cost of
//   row_shr:1 (inside 16-lane rows), wave_shr:1 (whole wave), v_fmac_f32_dpp (inline asm: the builtin's v_mov_b32_dpp is not
//   folded into v_fmac by hipcc 7.2), ds_bpermute_b32, v_permlane32_swap.
// Prints the lane image of each shift for lane-id input and ns per wave-instruction (all CUs busy, 4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
enum { ROW_SHR1 = 0, WAVE_SHR1 = 1, FMAC_DPP = 2, BPERMUTE = 3, MOV_FMAC = 4, NMODE = 5 };
template <int MODE>
__device__ __forceinline__ float step(float x, float t, float a) {
    if (MODE == ROW_SHR1) return a + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xF, 0xF, true));
    if (MODE == WAVE_SHR1) return a + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xF, 0xF, true));
    if (MODE == FMAC_DPP) {
        asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(x), "v"(t));
        return a;
    }
    if (MODE == BPERMUTE) return a + __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((((int)threadIdx.x & 63) - 1) * 4, __builtin_bit_cast(int, x)));
    return fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xF, 0xF, true)), t, a);
}
template <int MODE>
__global__ void image(float* out) {
    const float x = (float)(threadIdx.x & 63);
    out[threadIdx.x] = step<MODE>(x, 1.0f, 0.0f);
}
template <int MODE>
__global__ __launch_bounds__(256) void timed(float* out, int iters) {
    float x0 = threadIdx.x * 0.5f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, t = 1.0001f;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = step<MODE>(x0, t, a0); a1 = step<MODE>(x1, t, a1); a2 = step<MODE>(x2, t, a2); a3 = step<MODE>(x3, t, a3);
        }
        x0 += a3 * 1e-30f;   // keep the loop body live
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int MODE>
void run(const char* name, float* d, int per_step) {
    std::vector<float> h(64);
    image<MODE><<<1, 64>>>(d);
    hipMemcpy(h.data(), d, 64 * sizeof(float), hipMemcpyDeviceToHost);
    printf("%-10s lane image:", name);
    for (int l : {0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 63}) printf(" %d<-%g", l, h[l]);
    const int iters = 2000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    timed<MODE><<<blocks, 256>>>(d, 10);
    hipEventRecord(e0);
    timed<MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // 4 waves per block -> one per SIMD; 4 blocks per CU -> 4 waves per SIMD; per SIMD: 4 waves x iters x 32 steps
    printf("   %.2f ns per step and SIMD (%d instr per step)\n", ms * 1e6 / (4.0 * iters * 32), per_step);
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    run<ROW_SHR1>("row_shr:1", d, 1);
    run<WAVE_SHR1>("wave_shr:1", d, 1);
    run<FMAC_DPP>("fmac_dpp", d, 2);
    run<MOV_FMAC>("mov+fmac", d, 2);
    run<BPERMUTE>("bpermute", d, 2);
    return 0;
}
