// Development probe: v_cndmask_b32 forms (mask in VCC vs an SGPR pair; constant vs register sources).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ __launch_bounds__(1024) void probe(int iters, long long* out, float seed, unsigned long long m) {
    const int lane = threadIdx.x & 63;
    float r0 = seed + lane, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7, a = 1.5f, b = 2.5f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define C8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
#define A(k) "v_cndmask_b32_e32 %" #k ", %8, %9, vcc\n"
#define B(k) "v_cndmask_b32_e64 %" #k ", %8, %9, %10\n"
#define C(k) "v_cndmask_b32_e64 %" #k ", %" #k ", 0, %10\n"
#define D(k) "v_mov_b32 %" #k ", %8\n"
#define E(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %9, %10\n"
        if (OP == 0) asm volatile(C8(A) C8(A) C8(A) C8(A) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "s"(m) : "vcc");
        if (OP == 1) asm volatile(C8(B) C8(B) C8(B) C8(B) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "s"(m));
        if (OP == 2) asm volatile(C8(C) C8(C) C8(C) C8(C) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "s"(m));
        if (OP == 3) asm volatile(C8(D) C8(D) C8(D) C8(D) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "s"(m));
        if (OP == 4) asm volatile(C8(E) C8(E) C8(E) C8(E) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "s"(m));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if (s == 12345.678f) out[0] = 1;
    if (lane == 0) out[1 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int OP> void run(const char* name, long long* dbuf) {
    for (int w : {4, 8, 16}) {
        const int iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        probe<OP><<<256, 64 * w>>>(iters / 8, dbuf, 1.0f, 0x5555555555555555ull);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<OP><<<256, 64 * w>>>(iters, dbuf, 1.0f, 0x5555555555555555ull);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(1 + 256 * w);
        hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (size_t i = 1; i < h.size(); ++i) sum += h[i];
        const double n = iters * 32.0;
        printf("%-44s waves/SIMD %d: %6.2f ticks/instr/wave  %6.2f ns/instr/SIMD\n", name, w / 4, sum / (h.size() - 1) / n, ms * 1e6 / n / (w / 4));
    }
}
int main() {
    long long* dbuf; hipMalloc(&dbuf, 8 * (1 + 256 * 16));
    run<0>("v_cndmask_b32_e32 d, a, b, vcc", dbuf);
    run<1>("v_cndmask_b32_e64 d, a, b, s[pair]", dbuf);
    run<2>("v_cndmask_b32_e64 d, d, 0, s[pair]", dbuf);
    run<3>("v_mov_b32 d, a", dbuf);
    run<4>("v_cndmask_b32_e64 d, d, b, s[pair]", dbuf);
    return 0;
}
