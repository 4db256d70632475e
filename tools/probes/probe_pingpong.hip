// Development probe: can the mel decoder's K loop (ds_read_b128 + v_mfma_f32_32x32x16_f16) of one set of waves run under the
// VALU / LDS phases (bias+tanh+store, LayerNorm-like, depthwise-like) of ANOTHER set of waves on the same SIMDs?
// One 1024-thread workgroup per CU: waves 0-7 = group A (two per SIMD), waves 8-15 = group B (two per SIMD).
//   hipcc --offload-arch=gfx950 -O2 -w -o probe_pingpong probe_pingpong.hip && ./probe_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
enum { IDLE = 0, KLOOP = 1, TANH = 2, LNORM = 3, DWCONV = 4 };

__device__ __forceinline__ float dpp_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_hm(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_m(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)); }

template <int ROLE>
__device__ __forceinline__ float work(int iters, float* tile, int tid) {   // tid: 0..511 within the group; tile: [132][132] floats
    const int lane = tid & 63, w = tid >> 6;
    float s = 0.f;
    if (ROLE == KLOOP) {   // one "layer": 8 k-steps x 2 row tiles x (2 ds_read_b128 + 3 MFMA)
        f32x16 a0 = {0}, a1 = {0};
        const unsigned* base = reinterpret_cast<const unsigned*>(tile) + ((2 + 64 * (w >> 2) + (lane & 31)) * 132 + 4 * (lane >> 5));
        u32x4 wa = {1, 2, 3, 4}, wb = {5, 6, 7, 8};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int st = 0; st < 8; ++st) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const unsigned* ap = base + 32 * mt * 132 + 8 * st;
                    const u32x4 h1 = *reinterpret_cast<const u32x4*>(ap), h2 = *reinterpret_cast<const u32x4*>(ap + 64);
                    f32x16& acc = mt ? a1 : a0;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa), __builtin_bit_cast(f16x8, h2), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wb), __builtin_bit_cast(f16x8, h1), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa), __builtin_bit_cast(f16x8, h1), acc, 0, 0, 0);
                }
            }
            wa[0] += 1;
        }
        for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    } else if (ROLE == TANH) {   // 32 elements per lane: fma, exp, add, rcp, fma; 8 ds_write_b128
        float* base = tile + ((2 + 64 * (w >> 2) + (lane & 31)) * 132 + 32 * (w & 3) + 4 * (lane >> 5));
        f32x4 v[8];
        for (int q = 0; q < 8; ++q) v[q] = f32x4{0.1f * lane, 0.2f, 0.3f * q, 0.4f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(v[q][e], 0.7f, 0.01f * it)));
                *reinterpret_cast<f32x4*>(base + (q >> 2) * 32 * 132 + 8 * (q & 3)) = o;
                v[q] = o;
            }
        }
        for (int q = 0; q < 8; ++q) s += v[q][0] + v[q][3];
    } else if (ROLE == LNORM) {   // 16 lanes per row, 4 rows per thread, 8 channels each: load, 2 DPP reductions, normalise, store
        float* base = tile + ((2 + 16 * w + 4 * (lane >> 4)) * 132 + 4 * (lane & 15));
        for (int it = 0; it < iters; ++it) {
            f32x4 v[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j][0] = *reinterpret_cast<const f32x4*>(base + j * 132); v[j][1] = *reinterpret_cast<const f32x4*>(base + j * 132 + 64); }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sm = (v[j][0][0] + v[j][0][1]) + (v[j][0][2] + v[j][0][3]) + (v[j][1][0] + v[j][1][1]) + (v[j][1][2] + v[j][1][3]);
                sm += dpp_xor1(sm); sm += dpp_xor2(sm); sm += dpp_hm(sm); sm += dpp_m(sm);
                const float mean = sm * (1.0f / 128);
                float q = 0.f;
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[j][k][e] - mean; q = fmaf(d, d, q); }
                q += dpp_xor1(q); q += dpp_xor2(q); q += dpp_hm(q); q += dpp_m(q);
                const float rstd = __builtin_amdgcn_rsqf(q * (1.0f / 128) + 1e-5f);
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[j][k][e] = fmaf((v[j][k][e] - mean) * rstd, 1.01f, 0.02f);
                *reinterpret_cast<f32x4*>(base + j * 132) = v[j][0];
                *reinterpret_cast<f32x4*>(base + j * 132 + 64) = v[j][1];
            }
        }
        s = base[0];
    } else if (ROLE == DWCONV) {   // 4 channels x 8 rows per thread: 12 row loads, 5-tap fma, split to f16 planes, 16 ds_write_b64
        float* col = tile + ((8 * (tid >> 5)) * 132 + 4 * (tid & 31));
        const f32x4 tp[5] = {f32x4{.1f, .2f, .3f, .4f}, f32x4{.2f, .1f, .3f, .1f}, f32x4{.5f, .4f, .3f, .2f}, f32x4{.1f, .1f, .2f, .2f}, f32x4{.3f, .2f, .1f, .1f}};
        for (int it = 0; it < iters; ++it) {
            f32x4 win[12];
#pragma unroll
            for (int r = 0; r < 12; ++r) win[r] = *reinterpret_cast<const f32x4*>(col + r * 132);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                f32x4 a = {0.01f, 0.01f, 0.01f, 0.01f};
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = fmaf(win[r + j][e], tp[j][e], a[e]);
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const auto h0 = __builtin_amdgcn_cvt_pkrtz(a[0], a[1]);
                const auto h1 = __builtin_amdgcn_cvt_pkrtz(a[2], a[3]);
                const auto g0 = __builtin_amdgcn_cvt_pkrtz(a[0] - (float)h0[0], a[1] - (float)h0[1]);
                const auto g1 = __builtin_amdgcn_cvt_pkrtz(a[2] - (float)h1[0], a[3] - (float)h1[1]);
                unsigned* rowp = reinterpret_cast<unsigned*>(col + (r + 2) * 132) - 2 * (tid & 31);
                *reinterpret_cast<u32x2*>(rowp) = u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
                *reinterpret_cast<u32x2*>(rowp + 64) = u32x2{__builtin_bit_cast(unsigned, g0), __builtin_bit_cast(unsigned, g1)};
            }
        }
        s = col[0];
    }
    return s;
}

template <int RA, int RB>
__global__ __launch_bounds__(1024) void probe(int iters, long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int e = threadIdx.x; e < 2 * 132 * 132; e += blockDim.x) lds[e] = 0.001f * (e & 1023);
    __syncthreads();
    const int g = threadIdx.x >> 9, tid = threadIdx.x & 511;
    float* tile = lds + g * 132 * 132;
    const long long t0 = __builtin_amdgcn_s_memtime();
    float s = g == 0 ? work<RA>(iters, tile, tid) : work<RB>(iters, tile, tid);
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (s == 1234.5f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int RA, int RB>
void run(const char* name, int iters, long long* dbuf, float* sink) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<RA, RB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int lds = 2 * 132 * 132 * 4;
    probe<RA, RB><<<256, 1024, lds>>>(iters / 4, dbuf, sink);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<RA, RB><<<256, 1024, lds>>>(iters, dbuf, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256 * 16);
    hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < 256; ++i) for (int w = 0; w < 16; ++w) (w < 8 ? a : b) += h[i * 16 + w];
    printf("%-28s wall %7.1f us  per iteration: A %8.0f cycles  B %8.0f cycles  (wall/iter %6.2f us)\n", name, ms * 1e3, a / (256 * 8) / iters, b / (256 * 8) / iters, ms * 1e3 / iters);
}

int main() {
    long long* dbuf; float* sink;
    hipMalloc(&dbuf, 8 * 256 * 16); hipMalloc(&sink, 64);
    const int it = 400;
    run<KLOOP, IDLE>("K alone", it, dbuf, sink);
    run<TANH, IDLE>("tanh alone", it, dbuf, sink);
    run<LNORM, IDLE>("LN alone", it, dbuf, sink);
    run<DWCONV, IDLE>("dw alone", it, dbuf, sink);
    run<KLOOP, KLOOP>("K + K", it, dbuf, sink);
    run<KLOOP, TANH>("K + tanh", it, dbuf, sink);
    run<KLOOP, LNORM>("K + LN", it, dbuf, sink);
    run<KLOOP, DWCONV>("K + dw", it, dbuf, sink);
    run<TANH, DWCONV>("tanh + dw", it, dbuf, sink);
    run<TANH, LNORM>("tanh + LN", it, dbuf, sink);
    run<TANH, TANH>("tanh + tanh", it, dbuf, sink);
    run<LNORM, DWCONV>("LN + dw", it, dbuf, sink);
    return 0;
}
