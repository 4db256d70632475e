// Calibration of rocprofv3's WRITE_SIZE / FETCH_SIZE on gfx950 against known byte counts, in the mel decoder's own store pattern
// (MI355X_MICROARCH.md: "WRITE_SIZE is uncalibrated -- calibrate on a known byte count in your own access pattern").
//   stream_store : 62.9 MB as one float4 per lane, fully coalesced
//   mel_store    : the same 62.9 MB as (196608 rows x 80 floats) written like mel_decoder_kernel's epilogue: wave (mh, ns),
//                  lane (i, h) stores float4 at row 64 mh + 32 mt + i, column 32 ns + 8 g + 4 h  (g = 0..3, columns < 80)
//   stream_load  : reads the 62.9 MB back, 16 B per lane
// run: rocprofv3 --pmc WRITE_SIZE --kernel-trace -- tools/probes/probe_write_size   (and --pmc FETCH_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr long ROWS = 196608, COLS = 80;
__global__ __launch_bounds__(256) void stream_store(f4* p, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) p[i] = f4{1.f, 2.f, 3.f, (float)i};
}
__global__ __launch_bounds__(512) void mel_store(float* p) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5, mh = w >> 2, ns = w & 3;
    const long row0 = (long)blockIdx.x * 128;
    for (int mt = 0; mt < 2; ++mt)
        for (int g = 0; g < 4; ++g) {
            const int col = 32 * ns + 8 * g + 4 * h;
            if (col >= COLS) continue;
            const long f = row0 + 64 * mh + 32 * mt + i;
            *reinterpret_cast<f4*>(p + f * COLS + col) = f4{1.f, 2.f, (float)col, (float)f};
        }
}
__global__ __launch_bounds__(256) void stream_load(const f4* p, long n4, float* out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    f4 v = i < n4 ? p[i] : f4{0, 0, 0, 0};
    if (v[0] + v[1] + v[2] + v[3] == -12345.f) out[0] = 1.f;
}
int main() {
    const long bytes = ROWS * COLS * 4, n4 = bytes / 16;
    float *a, *o; hipMalloc(&a, bytes); hipMalloc(&o, 4);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(stream_store, dim3((n4 + 255) / 256), dim3(256), 0, 0, (f4*)a, n4);
        hipLaunchKernelGGL(mel_store, dim3(ROWS / 128), dim3(512), 0, 0, a);
        hipLaunchKernelGGL(stream_load, dim3((n4 + 255) / 256), dim3(256), 0, 0, (const f4*)a, n4, o);
    }
    hipDeviceSynchronize();
    printf("bytes per kernel: %ld\n", bytes);
    return 0;
}
