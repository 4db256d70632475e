// Development probe: issue cost of the instruction classes the mel decoder's non-MFMA phases are made of, per wave and per SIMD,
// at 1 / 2 / 4 waves per SIMD (one workgroup per CU).  Every test is a loop around one `asm volatile` block of 32 independent
// instructions of one kind, timed with s_memtime inside the wave and with HIP events around the launch.
//   hipcc --offload-arch=gfx950 -O2 -w -o probe_valu probe_valu.hip && ./probe_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum { FMA = 0, PKFMA, EXP, RCP, CVTPK, DPPADD, CNDMASK, FMAMIX, DSR128, DSW128, DSW64, DSW32, MFMA16, MFMA_F4, MFMA_F6, MFMA_E2, MFMA_R128, CVTF32, PKADD, PKMUL, MOV, MFMA_DEP, MFMA_DEP2, NOPS };
static const char* kNames[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_rcp_f32", "v_cvt_pkrtz_f16_f32", "v_add_f32_dpp", "v_cndmask_b32",
                               "v_fma_mix_f32", "ds_read_b128", "ds_write_b128", "ds_write_b64", "ds_write_b32", "mfma_32x32x16_f16",
                               "mfma + 4 v_fma", "mfma + 6 v_fma", "mfma + 2 v_exp", "mfma + 1 ds_read_b128", "v_cvt_f32_f16", "v_pk_add_f32", "v_pk_mul_f32", "v_mov_b32", "mfma same acc x8", "mfma 2 accs aabb"};
// instructions counted per asm block (the thing the printed cycles are divided by)
static const int kPerBlock[] = {32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 8, 8, 8, 8, 8, 32, 32, 32, 32, 8, 8};

#define R8(op, a, b) op " %0, " a ", " b "\n" op " %1, " a ", " b "\n" op " %2, " a ", " b "\n" op " %3, " a ", " b "\n" \
                     op " %4, " a ", " b "\n" op " %5, " a ", " b "\n" op " %6, " a ", " b "\n" op " %7, " a ", " b "\n"

template <int OP>
__global__ __launch_bounds__(1024) void probe(int iters, long long* out, float seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    float r0 = seed + lane, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float a = 1.0001f, b = 0.5f;
    f32x2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7}, p4 = {r0, r2}, p5 = {r1, r3}, p6 = {r4, r6}, p7 = {r5, r7};
    f32x2 pa = {a, a}, pb = {b, b};
    f32x4 q0 = {r0, r1, r2, r3}, q1 = q0, q2 = q0, q3 = q0;
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    f16x8 ha = {1, 2, 3, 4, 5, 6, 7, 8}, hb = {1, 1, 1, 1, 1, 1, 1, 1};
    // conflict-free 16-byte accesses: lane stride 16 B + wave base
    const unsigned addr = (unsigned)(threadIdx.x * 16) % (32 * 1024);
    for (int e = threadIdx.x; e < 16 * 1024; e += blockDim.x) lds[e] = e;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (OP == FMA) {
            asm volatile(R8("v_fma_f32", "%8, %9", "%0") R8("v_fma_f32", "%8, %9", "%0") R8("v_fma_f32", "%8, %9", "%0") R8("v_fma_f32", "%8, %9", "%0")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (OP == MOV) {
            asm volatile(R8("v_add_f32", "%8", "%9") R8("v_add_f32", "%8", "%9") R8("v_add_f32", "%8", "%9") R8("v_add_f32", "%8", "%9")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (OP == PKFMA) {
            asm volatile(R8("v_pk_fma_f32", "%8, %9", "%0") R8("v_pk_fma_f32", "%8, %9", "%0") R8("v_pk_fma_f32", "%8, %9", "%0") R8("v_pk_fma_f32", "%8, %9", "%0")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));
        } else if (OP == PKADD) {
            asm volatile(R8("v_pk_add_f32", "%8", "%9") R8("v_pk_add_f32", "%8", "%9") R8("v_pk_add_f32", "%8", "%9") R8("v_pk_add_f32", "%8", "%9")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));
        } else if (OP == PKMUL) {
            asm volatile(R8("v_pk_mul_f32", "%8", "%9") R8("v_pk_mul_f32", "%8", "%9") R8("v_pk_mul_f32", "%8", "%9") R8("v_pk_mul_f32", "%8", "%9")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));
        } else if (OP == EXP) {
#define T8(op) op " %0, %8\n" op " %1, %8\n" op " %2, %8\n" op " %3, %8\n" op " %4, %8\n" op " %5, %8\n" op " %6, %8\n" op " %7, %8\n"
            asm volatile(T8("v_exp_f32") T8("v_exp_f32") T8("v_exp_f32") T8("v_exp_f32")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));
        } else if (OP == RCP) {
            asm volatile(T8("v_rcp_f32") T8("v_rcp_f32") T8("v_rcp_f32") T8("v_rcp_f32")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
        } else if (OP == CVTF32) {
            asm volatile(T8("v_cvt_f32_f16") T8("v_cvt_f32_f16") T8("v_cvt_f32_f16") T8("v_cvt_f32_f16")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
        } else if (OP == CVTPK) {
            asm volatile(R8("v_cvt_pkrtz_f16_f32", "%8", "%9") R8("v_cvt_pkrtz_f16_f32", "%8", "%9") R8("v_cvt_pkrtz_f16_f32", "%8", "%9") R8("v_cvt_pkrtz_f16_f32", "%8", "%9")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (OP == DPPADD) {
#define D8 "v_add_f32_dpp %0, %8, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %1, %8, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" \
           "v_add_f32_dpp %2, %8, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %3, %8, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" \
           "v_add_f32_dpp %4, %8, %4 row_mirror row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %5, %8, %5 row_mirror row_mask:0xf bank_mask:0xf\n" \
           "v_add_f32_dpp %6, %8, %6 row_half_mirror row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %7, %8, %7 row_half_mirror row_mask:0xf bank_mask:0xf\n"
            asm volatile(D8 D8 D8 D8 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
        } else if (OP == CNDMASK) {
            asm volatile(R8("v_cndmask_b32", "%8, %9", "vcc") R8("v_cndmask_b32", "%8, %9", "vcc") R8("v_cndmask_b32", "%8, %9", "vcc") R8("v_cndmask_b32", "%8, %9", "vcc")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
        } else if (OP == FMAMIX) {
#define M8 "v_fma_mix_f32 %0, %8, -1.0, %0 op_sel_hi:[1,0,0]\n" "v_fma_mix_f32 %1, %8, -1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n" \
           "v_fma_mix_f32 %2, %8, -1.0, %2 op_sel_hi:[1,0,0]\n" "v_fma_mix_f32 %3, %8, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n" \
           "v_fma_mix_f32 %4, %8, -1.0, %4 op_sel_hi:[1,0,0]\n" "v_fma_mix_f32 %5, %8, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n" \
           "v_fma_mix_f32 %6, %8, -1.0, %6 op_sel_hi:[1,0,0]\n" "v_fma_mix_f32 %7, %8, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
            asm volatile(M8 M8 M8 M8 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
        } else if (OP == DSR128) {
#define L4(o) "ds_read_b128 %0, %4 offset:" #o "\n" "ds_read_b128 %1, %4 offset:" #o "+16384\n" "ds_read_b128 %2, %4 offset:" #o "+32768\n" "ds_read_b128 %3, %4 offset:" #o "+49152\n"
            asm volatile(L4(0) L4(0) L4(0) L4(0) L4(0) L4(0) L4(0) L4(0) "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(addr) : "memory");
        } else if (OP == DSW128) {
#define W4(ins) ins " %4, %0\n" ins " %4, %1 offset:16384\n" ins " %4, %2 offset:32768\n" ins " %4, %3 offset:49152\n"
            asm volatile(W4("ds_write_b128") W4("ds_write_b128") W4("ds_write_b128") W4("ds_write_b128") W4("ds_write_b128") W4("ds_write_b128") W4("ds_write_b128") W4("ds_write_b128") "s_waitcnt lgkmcnt(0)\n"
                         :: "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(addr) : "memory");
        } else if (OP == DSW64) {
            asm volatile(W4("ds_write_b64") W4("ds_write_b64") W4("ds_write_b64") W4("ds_write_b64") W4("ds_write_b64") W4("ds_write_b64") W4("ds_write_b64") W4("ds_write_b64") "s_waitcnt lgkmcnt(0)\n"
                         :: "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(addr) : "memory");
        } else if (OP == DSW32) {
            asm volatile(W4("ds_write_b32") W4("ds_write_b32") W4("ds_write_b32") W4("ds_write_b32") W4("ds_write_b32") W4("ds_write_b32") W4("ds_write_b32") W4("ds_write_b32") "s_waitcnt lgkmcnt(0)\n"
                         :: "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(addr) : "memory");
        } else {   // MFMA-based blocks: 8 MFMAs on 4 accumulators, with fillers in between
#define MF(c) "v_mfma_f32_32x32x16_f16 %" #c ", %12, %13, %" #c "\n"
#define F1(k) "v_fma_f32 %" #k ", %14, %15, %" #k "\n"
#define E1(k) "v_exp_f32 %" #k ", %" #k "\n"
            if (OP == MFMA_DEP)
                asm volatile(MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0)
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                             : "v"(ha), "v"(hb), "v"(a), "v"(b));
            else if (OP == MFMA_DEP2)
                asm volatile(MF(0) MF(0) MF(0) MF(1) MF(1) MF(1) MF(0) MF(1)
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                             : "v"(ha), "v"(hb), "v"(a), "v"(b));
            else if (OP == MFMA16)
                asm volatile(MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3)
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                             : "v"(ha), "v"(hb), "v"(a), "v"(b));
            else if (OP == MFMA_F4)
                asm volatile(MF(0) F1(4) F1(5) F1(6) F1(7) MF(1) F1(8) F1(9) F1(10) F1(11) MF(2) F1(4) F1(5) F1(6) F1(7) MF(3) F1(8) F1(9) F1(10) F1(11)
                             MF(0) F1(4) F1(5) F1(6) F1(7) MF(1) F1(8) F1(9) F1(10) F1(11) MF(2) F1(4) F1(5) F1(6) F1(7) MF(3) F1(8) F1(9) F1(10) F1(11)
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                             : "v"(ha), "v"(hb), "v"(a), "v"(b));
            else if (OP == MFMA_F6)
                asm volatile(MF(0) F1(4) F1(5) F1(6) F1(7) F1(8) F1(9) MF(1) F1(10) F1(11) F1(4) F1(5) F1(6) F1(7) MF(2) F1(8) F1(9) F1(10) F1(11) F1(4) F1(5) MF(3) F1(6) F1(7) F1(8) F1(9) F1(10) F1(11)
                             MF(0) F1(4) F1(5) F1(6) F1(7) F1(8) F1(9) MF(1) F1(10) F1(11) F1(4) F1(5) F1(6) F1(7) MF(2) F1(8) F1(9) F1(10) F1(11) F1(4) F1(5) MF(3) F1(6) F1(7) F1(8) F1(9) F1(10) F1(11)
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                             : "v"(ha), "v"(hb), "v"(a), "v"(b));
            else if (OP == MFMA_E2)
                asm volatile(MF(0) E1(4) E1(5) MF(1) E1(6) E1(7) MF(2) E1(8) E1(9) MF(3) E1(10) E1(11) MF(0) E1(4) E1(5) MF(1) E1(6) E1(7) MF(2) E1(8) E1(9) MF(3) E1(10) E1(11)
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                             : "v"(ha), "v"(hb), "v"(a), "v"(b));
            else if (OP == MFMA_R128) {
#define LR(k) "ds_read_b128 %" #k ", %16\n"
                asm volatile(MF(0) LR(4) MF(1) LR(5) MF(2) LR(6) MF(3) LR(7) MF(0) LR(4) MF(1) LR(5) MF(2) LR(6) MF(3) LR(7) "s_waitcnt lgkmcnt(0)\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                             : "v"(ha), "v"(hb), "v"(a), "v"(b), "v"(addr) : "memory");
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[0] + p6[0] + p7[0] + q0[0] + q1[1] + q2[2] + q3[3];
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 12345.678f) out[0] = 1;   // keep everything alive
    if (lane == 0) out[1 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(int waves_per_wg, int iters, long long* dbuf) {
    const int grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    probe<OP><<<grid, 64 * waves_per_wg, 64 * 1024>>>(iters / 8, dbuf, 1.0f);   // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<OP><<<grid, 64 * waves_per_wg, 64 * 1024>>>(iters, dbuf, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(1 + grid * waves_per_wg);
    hipMemcpy(h.data(), dbuf, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    for (size_t i = 1; i < h.size(); ++i) { sum += h[i]; mx = h[i] > mx ? h[i] : mx; }
    const double ticks = sum / (h.size() - 1);
    const double n = (double)iters * kPerBlock[OP];
    // ticks per instruction per wave; and per SIMD = / waves per SIMD
    printf("%-24s waves/SIMD %d : %8.2f ticks/instr/wave  %7.2f ticks/instr/SIMD   wall %8.3f ms -> %7.2f ns/instr/SIMD  (max wave %.0f ticks)\n",
           kNames[OP], waves_per_wg / 4, ticks / n, ticks / n / (waves_per_wg / 4), ms, ms * 1e6 / n / (waves_per_wg / 4), mx);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int OP>
void run_all(long long* dbuf, int iters) {
    for (int w : {4, 8, 16}) run<OP>(w, iters, dbuf);
}

int main() {
    long long* dbuf;
    hipMalloc(&dbuf, sizeof(long long) * (1 + 256 * 16));
    const int it = 20000;
    run_all<MFMA_DEP>(dbuf, it); run_all<MFMA_DEP2>(dbuf, it); run_all<MFMA16>(dbuf, it);
    if (getenv("PROBE_MFMA_ONLY")) return 0;
    run_all<FMA>(dbuf, it); run_all<MOV>(dbuf, it); run_all<PKFMA>(dbuf, it); run_all<PKADD>(dbuf, it); run_all<PKMUL>(dbuf, it);
    run_all<EXP>(dbuf, it); run_all<RCP>(dbuf, it); run_all<CVTPK>(dbuf, it); run_all<CVTF32>(dbuf, it);
    run_all<DPPADD>(dbuf, it); run_all<CNDMASK>(dbuf, it); run_all<FMAMIX>(dbuf, it);
    run_all<DSR128>(dbuf, it / 4); run_all<DSW128>(dbuf, it / 4); run_all<DSW64>(dbuf, it / 4); run_all<DSW32>(dbuf, it / 4);
    run_all<MFMA16>(dbuf, it); run_all<MFMA_F4>(dbuf, it); run_all<MFMA_F6>(dbuf, it); run_all<MFMA_E2>(dbuf, it); run_all<MFMA_R128>(dbuf, it);
    hipFree(dbuf);
    return 0;
}
