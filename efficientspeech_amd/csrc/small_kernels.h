// Bandwidth-trivial kernels of the path: weight re-layout, mask pooling, the variance-adaptor
// tail (bucketize / embedding / concat / duration rounding), the length regulator scan and the
// optional materialisation of the upsampled features.
#pragma once
#include "esmi_dev.h"

namespace esmi {

// dst[j][o][i] = src[o][i][j]   (nn.Conv1d (Cout,Cin,k) -> tap-major)   when transposed == 0
// dst[j][o][i] = src[i][o][j]   (nn.ConvTranspose1d (Cin,Cout,k))       when transposed == 1
static __global__ void pack_conv_kernel(const float* __restrict__ src, float* __restrict__ dst, int cout, int cin, int k,
                                 int transposed, int* __restrict__ zero_slot = nullptr, int flip = 0) {
    if (zero_slot && blockIdx.x == 0 && threadIdx.x == 0) zero_slot[0] = 0;   // (training: the absmax slot of the GEMM this pack precedes)
    const long n = (long)cout * cin * k;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int i = (int)(e % cin);
        const int o = (int)((e / cin) % cout);
        const int j = (int)(e / ((long)cin * cout));
        const int js = flip ? k - 1 - j : j;     // flip: taps reversed (a stride-1 transposed convolution AS a plain one, padding k - 1 - pad)
        dst[e] = transposed ? src[((long)i * cout + o) * k + js] : src[((long)o * cin + i) * k + js];
    }
}

// MFMA B-fragment packing of `taps` row-major (N, K) matrices (wave_chain.h), NT = ceil(N/32):
//   dst[(((t*(K/8) + kc)*NT + nt)*64 + lane)*4 + s] = src[(t*N + 32*nt + (lane&31))*K + 8*kc + 4*(lane>>5) + s]   (0 for rows >= N)
static __global__ void pack_bfrag_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int K, int NT, int taps) {
    const long per = (long)(K / 8) * NT * 256;
    const long n = per * taps;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int t = (int)(e / per);
        const long f = e - (long)t * per;
        const int s = (int)(f & 3);
        const int lane = (int)((f >> 2) & 63);
        const int nt = (int)((f >> 8) % NT);
        const int kc = (int)((f >> 8) / NT);
        const int row = 32 * nt + (lane & 31);
#if ESMI_CHAIN_SPLIT
        // split-f16x2 operands (esmi_dev.h): within a group of four k-steps (32 channels) the four 1 KiB slots hold
        // {step 0 piece 1, step 0 piece 2, step 1 piece 1, step 1 piece 2}; a lane's 8 k-slots of 16-channel step st are
        // the channels 16*st + 4*(lane>>5) + (0..3) and + 8 of that -- exactly the two float4 the A side reads there
        // (wave_fetch_a), so A needs no shuffle.  Dword s = k-slots 2s, 2s + 1;  values are 2^8 * W.
        const int st = (kc >> 1) & 1, pl = kc & 1;
        const int ch0 = 32 * (kc >> 2) + 16 * st + 4 * (lane >> 5) + (s < 2 ? 2 * s : 8 + 2 * (s - 2));
        unsigned half[2];
        for (int j = 0; j < 2; ++j) {
            const float x = (row < N ? src[((long)t * N + row) * K + ch0 + j] : 0.0f) * kF16WScale;
            const unsigned h1 = f32_to_f16_bits(x, false);
            half[j] = pl == 0 ? h1 : f32_to_f16_bits(x - f16_bits_to_f32(h1), false);
        }
        dst[e] = __builtin_bit_cast(float, half[0] | (half[1] << 16));
#else
        const int colk = 8 * kc + 4 * (lane >> 5) + s;
        dst[e] = row < N ? src[((long)t * N + row) * K + colk] : 0.0f;
#endif
    }
}

// depthwise weight (C,1,k) -> tap-major (k, C)
static __global__ void pack_dw_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int k) {
    const int n = C * k;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int c = e % C, j = e / C;
        dst[e] = src[c * k + j];
    }
}

// copy n floats, zero-fill up to n_pad
static __global__ void copy_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int n_pad) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_pad; e += gridDim.x * blockDim.x)
        dst[e] = e < n ? src[e] : 0.0f;
}

// blocks.py:51-57: F.pad(mask, value=True) to a multiple of pool, max over groups of pool
static __global__ void pool_mask_kernel(const unsigned char* __restrict__ mask, int B, int T, int pool,
                                 unsigned char* __restrict__ out, int n_out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * n_out) return;
    const int b = e / n_out, n = e - b * n_out;
    unsigned char m = 0;
    for (int q = 0; q < pool; ++q) {
        const int t = n * pool + q;
        m |= (t >= T) ? (unsigned char)1 : mask[b * T + t];
    }
    out[e] = m;
}

// torch.bucketize(v, edges, right=False): number of edges strictly below v
__device__ __forceinline__ int bucketize_left(float v, const float* __restrict__ edges, int n) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (edges[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

struct VaTailP {
    int rows, T, dim;  // rows = B*T
    const unsigned char* mask;
    const float *pitch_pred, *energy_pred, *dur_pred;   // (rows)
    const float *pitch_t, *energy_t;                    // teacher values or NULL
    const int* dur_t;                                   // forced durations or NULL
    const float *pbins, *ebins, *pemb, *eemb;
    float* feat;  // (rows, 4*dim): channels [dim,2dim) pitch emb, [2dim,3dim) energy emb
    int *pitch_idx, *energy_idx, *dur;
};

// networks.py:349-384 minus the convolutions: one thread per (row, channel)
static __global__ void va_tail_kernel(const VaTailP p) {   // one thread = (row, four consecutive channels); dim % 4 == 0
    const int q4 = p.dim >> 2;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)p.rows * q4) return;
    const int row = (int)(e / q4), c = 4 * (int)(e - (long)row * q4);
    const bool pad = p.mask && p.mask[row];
    const int pi = bucketize_left(p.pitch_t ? p.pitch_t[row] : p.pitch_pred[row], p.pbins, p.dim - 1);
    const int ei = bucketize_left(p.energy_t ? p.energy_t[row] : p.energy_pred[row], p.ebins, p.dim - 1);
    float* fr = p.feat + (long)row * 4 * p.dim;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(fr + p.dim + c) = pad ? z : *reinterpret_cast<const f32x4*>(p.pemb + (long)pi * p.dim + c);
    *reinterpret_cast<f32x4*>(fr + 2 * p.dim + c) = pad ? z : *reinterpret_cast<const f32x4*>(p.eemb + (long)ei * p.dim + c);
    if (c == 0) {
        if (p.pitch_idx) p.pitch_idx[row] = pi;
        if (p.energy_idx) p.energy_idx[row] = ei;
        float d = p.dur_t ? (float)p.dur_t[row] : rintf(p.dur_pred[row]);  // torch.round: half to even
        if (p.mask) {                                                      // networks.py:381-382
            if (pad) d = 0.0f;
            d = fmaxf(d, 0.0f);
        }
        p.dur[row] = (int)d;  // FeatureUpsampler `.int()`, networks.py:234
    }
}

// networks.py:233-244 as a scan: one wave per utterance, inclusive cumsum of max(dur,0)
static __global__ __launch_bounds__(64) void length_regulate_kernel(const int* __restrict__ dur, int T, int* __restrict__ cum,
                                                             int* __restrict__ mel_len, int* __restrict__ lmax) {
    const int b = blockIdx.x, lane = lane_id();
    const int per = (T + 63) / 64;
    const int t0 = lane * per;
    int local = 0;
    for (int q = 0; q < per; ++q) {
        const int t = t0 + q;
        if (t < T) local += max(dur[b * T + t], 0);
    }
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = shfl_up_i(incl, d);
        if (lane >= d) incl += v;
    }
    int run = incl - local;
    for (int q = 0; q < per; ++q) {
        const int t = t0 + q;
        if (t < T) {
            run += max(dur[b * T + t], 0);
            cum[b * T + t] = run;
        }
    }
    const int total = shfl_i(incl, 63);
    if (lane == 0) {
        mel_len[b] = total;
        if (lmax) atomicMax(lmax, total);
    }
}

// out[0] = max(v[0..n), 0): one workgroup, no atomics, no zero-initialised output needed
static __global__ __launch_bounds__(64) void max_i32_kernel(const int* __restrict__ v, int n, int* __restrict__ out) {
    const int lane = lane_id();
    int m = 0;
    for (int j = lane; j < n; j += 64) m = max(m, v[j]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, shfl_i(m, lane ^ d));
    if (lane == 0) out[0] = m;
}

// first i with cum[i] > f  (searchsorted right); T if none
__device__ __forceinline__ int frame_to_phoneme(const int* __restrict__ cum, int T, int f) {
    int lo = 0, hi = T;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cum[mid] > f) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

static __global__ void lr_indices_kernel(const int* __restrict__ cum, int B, int T, int L, int* __restrict__ idx) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)B * L) return;
    const int b = (int)(e / L), f = (int)(e - (long)b * L);
    const int i = frame_to_phoneme(cum + b * T, T, f);
    idx[e] = i < T ? i : -1;
}

// FeatureUpsampler output (networks.py:246-255): features (B,L,C) and masks (B,L); one float4 per thread
static __global__ void upsample_kernel(const float* __restrict__ feat, const unsigned char* __restrict__ fmask,
                                const int* __restrict__ cum, int B, int T, int C, int L, float* __restrict__ out,
                                unsigned char* __restrict__ omask) {
    const int c4 = C >> 2;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)B * L * c4) return;
    const int q = (int)(e % c4);
    const long bf = e / c4;
    const int b = (int)(bf / L), f = (int)(bf - (long)b * L);
    const int i = frame_to_phoneme(cum + b * T, T, f);
    f32x4 v = zero4();
    if (i < T) v = ld4(feat + ((long)b * T + i) * C + 4 * q);
    *reinterpret_cast<f32x4*>(out + bf * C + 4 * q) = v;
    if (q == 0 && omask) omask[bf] = i < T ? (fmask ? fmask[b * T + i] : (unsigned char)0) : (unsigned char)1;
}

// get_embedding, networks.py:128-149: idx = bucketize(v, bins) (right = False), out row = emb[idx]; one thread per (row, channel)
static __global__ void bucket_embed_kernel(const float* __restrict__ v, const float* __restrict__ bins, const float* __restrict__ emb,
                                    long rows, int dim, float* __restrict__ out, int* __restrict__ idx) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * dim) return;
    const long row = e / dim;
    const int c = (int)(e - row * dim);
    const int bi = bucketize_left(v[row], bins, dim - 1);
    out[e] = emb[(long)bi * dim + c];
    if (c == 0 && idx) idx[row] = bi;
}

// out[0] = max(out[0], max_i |x[i]|) as the bit pattern of a non-negative float (NaN counts as +inf); out[0] zeroed by the caller
static __global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, int* __restrict__ out) {
    int m = 0;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const float a = fabsf(x[e]);
        const int bits = a != a ? 0x7F800000 : (__builtin_bit_cast(int, a) & 0x7FFFFFFF);
        m = max(m, bits);
    }
    const int lane = lane_id();
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, shfl_i(m, lane ^ d));
    if (lane == 0 && m > 0) atomicMax(out, m);
}

static __global__ void mask_rows_kernel(float* __restrict__ x, const unsigned char* __restrict__ mask, long rows, int C) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * C) return;
    if (mask[e / C]) x[e] = 0.0f;
}

}  // namespace esmi
