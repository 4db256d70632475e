// Probe: which workgroups share a CU, and what identifies the second one?  Each workgroup (512 threads, 76 KB LDS: two per CU, as
// mel_decoder_kernel<128,5,8>) records HW_REG_LDS_ALLOC, HW_REG_HW_ID, XCC_ID and its start time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512, 4) void k(unsigned long long* out, int spin) {
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[4 * blockIdx.x + 0] = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11));   // HW_REG_LDS_ALLOC, all 32 bits
        out[4 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
        out[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
        out[4 * blockIdx.x + 3] = t0;
    }
    lds[threadIdx.x] = (float)t0;
    __syncthreads();
    float a = lds[(threadIdx.x + 1) & 511];
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) out[0] = 0;
}
int main() {
    const int n = 1792;
    unsigned long long* d; hipMalloc(&d, n * 32);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(n), dim3(512), 76 * 1024, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(n * 4); hipMemcpy(h.data(), d, n * 32, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull; for (int i = 0; i < n; ++i) tmin = h[4 * i + 3] < tmin ? h[4 * i + 3] : tmin;
    for (int i : {0, 1, 8, 16, 248, 255, 256, 257, 264, 504, 511, 512, 513, 520, 768, 1024, 1791})
        printf("wg %4d lds_alloc %08llx hw_id %08llx xcc %llx t0 %llu\n", i, h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3] - tmin);
    int nz = 0, first = 0; for (int i = 0; i < n; ++i) { nz += (h[4 * i] & 0xff) != 0; if (i < 512) first += (h[4 * i] & 0xff) != 0; }
    printf("lds_base != 0: %d of %d; among ids < 512: %d\n", nz, n, first);
    return 0;
}
