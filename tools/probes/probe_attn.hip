// Development probe: attn_lds_kernel<NKT, CK> at base ES's two attention shapes (per-op plan: q = x (Wq^T Wk) per head, keys = values = x
// shared by the heads), timed with events.  Build one binary per knock-out mask (ESMI_ATTN_KO, attention.h):
//   for ko in 0 1 2 4 8 16 32 64; do hipcc --offload-arch=gfx950 -O2 -std=c++17 -DESMI_ATTN_KO=$ko -o pa_$ko probe_attn.hip; done
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../efficientspeech_amd/csrc/attention.h"
using namespace esmi;
template <int NKT, int CK> static void run(int B, int N, int C, int h) {
    const size_t nx = (size_t)B * N * C, nq = nx * h;
    float *x, *q, *ctx;
    hipMalloc(&x, nx * 4); hipMalloc(&q, nq * 4); hipMalloc(&ctx, nq * 4);
    std::vector<float> hx(nx), hq(nq);
    srand(1);
    for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hq) v = (rand() % 2001 - 1000) * 1e-3f;
    hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(q, hq.data(), nq * 4, hipMemcpyHostToDevice);
    AttnP p = {};
    p.q = q; p.k = x; p.v = x; p.ldq = h * C; p.ldk = p.ldv = C; p.hsq = C; p.hsk = p.hsv = 0;
    p.B = B; p.N = N; p.C = C; p.h = h; p.scale = 1.0f / sqrtf((float)(C / h)); p.ctx = ctx;
    const size_t lds = attn_lds_bytes(N, C);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_lds_kernel<NKT, CK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((attn_lds_kernel<NKT, CK>), dim3(B * h), dim3(64 * NKT), lds, 0, p);
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((attn_lds_kernel<NKT, CK>), dim3(B * h), dim3(64 * NKT), lds, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("KO=%3d attn_lds<%d,%d> B=%d N=%d C=%d h=%d lds=%zu: %.1f us (%s)\n", ESMI_ATTN_KO, NKT, CK, B, N, C, h, lds, ms * 1e3 / reps, hipGetErrorString(hipGetLastError()));
    hipFree(x); hipFree(q); hipFree(ctx);
}
int main() {
    run<4, 128>(512, 128, 256, 2);
    run<8, 128>(512, 256, 128, 2);
    return 0;
}
