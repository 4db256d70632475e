// Development probe: where does the dispatcher put 1024 single-wave workgroups (17 KB LDS each) on MI355X?
// Prints the histogram of waves per SIMD and the MFMA cycles per instruction each wave saw.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(64, 1) void probe(unsigned* hw, unsigned* xcc, long long* cyc, int iters) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 0.f;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x, y = 1.0f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        hw[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        cyc[blockIdx.x] = t1 - t0;
    }
    if (s == 12345.f) lds[0] = s;
}
int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 1024, ldsb = argc > 2 ? atoi(argv[2]) : 17408, iters = argc > 3 ? atoi(argv[3]) : 200;
    unsigned *hw, *xcc; long long* cyc;
    hipMalloc(&hw, n * 4); hipMalloc(&xcc, n * 4); hipMalloc(&cyc, n * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(n), dim3(64), ldsb, 0, hw, xcc, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned> h(n), x(n); std::vector<long long> c(n);
    hipMemcpy(h.data(), hw, n * 4, hipMemcpyDeviceToHost); hipMemcpy(x.data(), xcc, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), cyc, n * 8, hipMemcpyDeviceToHost);
    std::map<unsigned long long, int> per_simd, per_cu;
    for (int i = 0; i < n; ++i) {
        unsigned simd = (h[i] >> 4) & 3, cu = (h[i] >> 8) & 15, sh = (h[i] >> 12) & 1, se = (h[i] >> 13) & 7, xc = x[i] & 15;
        unsigned long long cuid = ((xc * 8 + se) * 2 + sh) * 16 + cu;
        per_cu[cuid]++; per_simd[cuid * 4 + simd]++;
    }
    std::map<int, int> hist_simd, hist_cu;
    for (auto& kv : per_simd) hist_simd[kv.second]++;
    for (auto& kv : per_cu) hist_cu[kv.second]++;
    printf("n=%d lds=%d: distinct CUs %zu, distinct SIMDs %zu\n", n, ldsb, per_cu.size(), per_simd.size());
    for (auto& kv : hist_cu) printf("  CUs with %d waves: %d\n", kv.first, kv.second);
    for (auto& kv : hist_simd) printf("  SIMDs with %d waves: %d\n", kv.first, kv.second);
    double tot = 0; long long mx = 0, mn = 1LL << 60;
    for (auto v : c) { tot += v; if (v > mx) mx = v; if (v < mn) mn = v; }
    printf("  cycles per MFMA: mean %.1f min %.1f max %.1f\n", tot / n / (4.0 * iters), mn / (4.0 * iters), mx / (4.0 * iters));
    printf("  first 16 (xcc,se,sh,cu,simd):");
    for (int i = 0; i < 16; ++i) printf(" (%u,%u,%u,%u,%u)", x[i] & 15, (h[i] >> 13) & 7, (h[i] >> 12) & 1, (h[i] >> 8) & 15, (h[i] >> 4) & 3);
    printf("\n");
    return 0;
}
