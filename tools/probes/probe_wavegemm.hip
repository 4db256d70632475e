// Development probe: cycles per MFMA of wave_gemm_taps<NT> in isolation (1 wave per SIMD), A rows and packed B from global.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../efficientspeech_amd/csrc/wave_chain.h"
using namespace esmi;
template <int NT, int KG>
__global__ __launch_bounds__(64, 1) void probe(const float* A, const float* Wp, float* out, long long* cyc, int ntaps) {
    constexpr int K = 32 * KG;
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    f32x16 acc[NT];
    zero_tiles<NT>(acc);
    const float* taps[5];
    bool tok[5];
    for (int j = 0; j < 5; ++j) { taps[j] = A + ((blockIdx.x * 37 + i + j) % 150) * K + 4 * h; tok[j] = (i + j) % 7 != 0; }
    long long t0 = __builtin_amdgcn_s_memtime();
    WaveGrp<NT> g0;
    wave_prefetch<NT>(g0, Wp, NT, 0, 0, lane);
    wave_gemm_taps<NT, 3, KG, true>(acc, g0, reinterpret_cast<const float* const (&)[3]>(taps), reinterpret_cast<const bool (&)[3]>(tok), Wp, (long)K * 32 * NT, NT, 0, 0, lane);
    float s = 0;
    for (int nt = 0; nt < NT; ++nt) for (int r = 0; r < 16; ++r) s += acc[nt][r];
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = s;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NT, int KG> void run(int n, int ntaps) {
    const int K = 32 * KG;
    float *A, *W, *out; long long* cyc;
    hipMalloc(&A, 150 * K * 4); hipMalloc(&W, (size_t)5 * K * 32 * NT * 4); hipMalloc(&out, n * 64 * 4); hipMalloc(&cyc, n * 8);
    hipMemset(A, 0, 150 * K * 4); hipMemset(W, 0, (size_t)5 * K * 32 * NT * 4);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<NT, KG>), dim3(n), dim3(64), 0, 0, A, W, out, cyc, ntaps);
    hipDeviceSynchronize();
    std::vector<long long> c(n);
    hipMemcpy(c.data(), cyc, n * 8, hipMemcpyDeviceToHost);
    double tot = 0; for (auto v : c) tot += v;
    const double mf = (double)ntaps * (K / 2) * NT;
    printf("NT=%d K=%d taps=%d waves=%d: %.0f cycles per wave, %.1f cycles per MFMA\n", NT, K, ntaps, n, tot / n, tot / n / mf);
}
int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 1024;
    run<4, 4>(n, 3); run<2, 4>(n, 3); run<1, 4>(n, 3); run<4, 4>(n, 1); run<1, 1>(n, 3); run<2, 2>(n, 3);
    return 0;
}
