// One HiFi-GAN ResBlock (hifigan/models.py:20-58 ResBlock1, :61-82 ResBlock2) as ONE kernel: the whole chain of
// dilated convolutions of a block runs on an LDS-resident window of the signal, so the stage input is read once and the
// block's result is written (or accumulated into the sum over the stage's ResBlocks, models.py:116-121) once -- instead of
// a global round trip per convolution (6 per ResBlock1), each of which re-read every input row k times.
//
//   ResBlock1:  for (c1, c2) in pairs:  xt = c1(lrelu(x)); xt = c2(lrelu(xt)); x = xt + x
//   ResBlock2:  for c in convs:         x = c(lrelu(x)) + x
//   both are a list of convs with a flag: add_res -> out = conv + bias + x, x <- out; otherwise out = conv + bias.
//
// Window: R rows (positions), channels-last.  A workgroup produces TL = R - 2.halo output positions, halo = the chain's
// one-sided receptive field sum((k-1)/2 . dil); every conv is evaluated on all R rows (rows whose inputs fall outside the
// window read clamped rows and hold finite garbage that never reaches the TL centre rows).  Rows outside the sequence
// [0, n) are forced to zero after every conv: that is the zero padding each Conv1d of the reference applies to ITS input.
//
// Arithmetic: split-f16x2 on the f16 matrix pipe (esmi_dev.h), both operands pre-split.  The activation is split once
// where it is produced -- leaky_relu(x) as two binary16 planes [row][C] in LDS -- and read as MFMA B fragments (one
// ds_read_b128 per plane: 8 channels of one tap-shifted row); the weights are pre-scaled by 2^8, pre-split and stored in
// A-fragment order by esmi_pack_resblock_f16 (one coalesced 1 KB read per plane and k-step, L1/L2 resident).
//   D[co][pos] += W[co][kidx] . X[kidx][pos],  kidx = tap . C + ci  (16 per MFMA: for C = 8 two taps per step)
// so a lane ends up holding 4 consecutive channels of one position: 8-byte plane writes, 16-byte global accesses.
// The residual stream x stays in registers in exactly that layout (fp32, never rounded).
//
// The planes are updated IN PLACE: a conv's K loop only reads them, its result waits in registers across a barrier (all reads
// done), is written, and a second barrier publishes it -- half the LDS of a ping-pong pair, so two workgroups share a CU and
// one's K loop (matrix pipe) runs under the other's epilogue (VALU) and barriers.
//
// Work split: 8 waves; wave = (pair of 32-row tiles, 32-channel M tile).  C < 32 pads M with zero weight rows (the MFMA
// work of the padding is wasted but the pipe is otherwise idle: these stages are the memory-bound ones).
#pragma once
#include "esmi_dev.h"

namespace esmi {

constexpr int kRbMaxConv = 6;
constexpr int kRbWaves = 8;

struct RbConv {
    const unsigned* wp;   // esmi_pack_resblock_f16: [mtile][step][plane 2][lane 64][4 dwords]
    const float* bias;    // (C)
    int dil, add_res;
};
struct ResblockP {
    const float* x;       // (B, n, C) stage input
    float* out;           // (B, n, C): out (+)= block(x)
    int B, n, k, n_conv, R, TL, halo, accum, tiles_per_b;
    float slope;
    RbConv conv[kRbMaxConv];
};

// per plane.  C >= 32 (32x32x16 tiles: lane = position, 32 consecutive rows per read): 16 pad bytes make the 16-byte reads
// conflict-free; C <= 16 (16x16x32 tiles: 16 consecutive rows, the k-blocks of a lane group 16 bytes or one tap apart): unpadded
__host__ __device__ constexpr int rb_row_bytes(int c) { return 2 * c + (c <= 16 ? 0 : 16); }
__host__ __device__ constexpr int rb_ksteps16(int c, int k) { return (c * k + 31) / 32; }
__host__ __device__ constexpr int rb_ksteps(int c, int k) { return (c * k + 15) / 16; }
__host__ __device__ constexpr int rb_mtiles(int c) { return c > 32 ? c / 32 : 1; }
__host__ __device__ inline size_t rb_pack_dwords(int c, int k) { return (size_t)(c <= 16 ? rb_ksteps16(c, k) : rb_mtiles(c) * rb_ksteps(c, k)) * 2 * 64 * 4; }
__host__ __device__ inline size_t rb_lds_bytes(int c, int R) { return (size_t)2 * R * rb_row_bytes(c); }

// (k, C, C) tap-major fp32 -> A fragments, scaled by 2^8, two nearest-rounded binary16 pieces
static __global__ void pack_resblock_kernel(const float* __restrict__ w, unsigned* __restrict__ dst, int c, int k) {
    const int steps = rb_ksteps(c, k);
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (mtile, step, lane)
    if (q >= (long)rb_mtiles(c) * steps * 64) return;
    const int lane = (int)(q & 63), s = (int)((q >> 6) % steps), m = (int)((q >> 6) / steps);
    const int co = 32 * m + (lane & 31), h = lane >> 5;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kidx = 16 * s + 8 * h + e, tap = kidx / c, ci = kidx - tap * c;
        v[e] = (co < c && tap < k) ? w[((long)tap * c + co) * c + ci] * kF16WScale : 0.0f;
    }
    unsigned* d = dst + ((long)(m * steps + s) * 2) * 256 + lane * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned h1, h2;
        split_f16_pair_rn(v[2 * j], v[2 * j + 1], h1, h2);
        d[j] = h1;
        d[256 + j] = h2;
    }
}

// the same for the 16x16x32 tiles of C <= 16: [step][plane 2][lane 64][4 dwords], lane = (co = lane & 15, k-block = lane >> 4)
static __global__ void pack_resblock16_kernel(const float* __restrict__ w, unsigned* __restrict__ dst, int c, int k) {
    const int steps = rb_ksteps16(c, k);
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (step, lane)
    if (q >= (long)steps * 64) return;
    const int lane = (int)(q & 63), s = (int)(q >> 6);
    const int co = lane & 15, kb = lane >> 4;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kidx = 32 * s + 8 * kb + e, tap = kidx / c, ci = kidx - tap * c;
        v[e] = (co < c && tap < k) ? w[((long)tap * c + co) * c + ci] * kF16WScale : 0.0f;
    }
    unsigned* d = dst + ((long)s * 2) * 256 + lane * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned h1, h2;
        split_f16_pair_rn(v[2 * j], v[2 * j + 1], h1, h2);
        d[j] = h1;
        d[256 + j] = h2;
    }
}

__device__ __forceinline__ f32x4 lrelu4(const f32x4& v, float slope) {   // 0 <= slope <= 1 (the launcher checks): max(v, slope v)
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaxf(v[e], v[e] * slope);
    return o;
}

#ifndef ESMI_RB_WPS     // (A/B knob; the product value is below)
#define ESMI_RB_WPS 3
#endif
constexpr int kRbWps = ESMI_RB_WPS;   // waves per SIMD the kernels are compiled for.  3 (round 6): 155-164 VGPRs, NO scratch, one 8-wave workgroup per CU:
                                      // 7.13-7.19 ms for the v2 generator at B = 32 against 7.27-7.29 ms at 4 (128 VGPRs, 88-116 B of scratch per lane, two
                                      // workgroups per CU) and 7.14-7.20 at 2 -- profiles/r06_probes/vocoder_wps.md; 6 / 8 spill more (4.1 / 5.2 vs 3.7 ms, round 2)
constexpr int kRbPd = 1;    // weight fragments are fetched this many k-steps ahead (deeper: spills at 128 VGPRs, 4.0-4.9 ms)

template <int C, int K>
__global__ __launch_bounds__(64 * kRbWaves, kRbWps) void hifigan_resblock_kernel(const ResblockP p) {
    constexpr int RS = rb_row_bytes(C), MT = rb_mtiles(C), CG = (C < 32 ? C : 32) / 8;
    constexpr int STEPS = rb_ksteps(C, K), HALF = (K - 1) / 2, PD = kRbPd < STEPS ? kRbPd : STEPS;
    static_assert(STEPS >= PD, "prefetch distance reaches at most into the next conv");
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    ESMI_DYN_LDS(lds_f);
    char* lds = reinterpret_cast<char*>(lds_f);   // ONE pair of planes [2][R][RS], updated in place between two barriers
    const int lane = lane_id(), w = wave_id(), i = lane & 31, h = lane >> 5;
    const int pair = w / MT, mt = w - pair * MT;
    const bool active = pair * 64 < p.R;
    const int b = (int)blockIdx.x / p.tiles_per_b;
    const int t0 = ((int)blockIdx.x - b * p.tiles_per_b) * p.TL - p.halo;   // sequence position of window row 0
    const int plane = p.R * RS;
    const int row0 = 64 * pair + i;
    const long wlane = (long)(mt * STEPS) * 512 + lane * 4;

    f32x4 xres[2][CG];
    bool inside[2];
    auto split4 = [&](const f32x4& v) __attribute__((always_inline)) {   // leaky_relu, then the two binary16 planes of 4 channels
        const f32x4 a = lrelu4(v, p.slope);
        unsigned h1a, h2a, h1b, h2b;
        split_f16_pair(a[0], a[1], h1a, h2a);
        split_f16_pair(a[2], a[3], h1b, h2b);
        return u32x4{h1a, h1b, h2a, h2b};
    };
    auto put_planes = [&](int row, int ch, const u32x4& v) __attribute__((always_inline)) {
        char* d = lds + row * RS + ch * 2;
        *reinterpret_cast<u32x2*>(d) = u32x2{v[0], v[1]};
        *reinterpret_cast<u32x2*>(d + plane) = u32x2{v[2], v[3]};
    };
    // weight fragments of flattened step index s (s >= STEPS: the next conv's first steps; past the last conv: nothing)
    auto wfetch = [&](int ci, int s, u32x4& hi, u32x4& lo) __attribute__((always_inline)) {
        if (s >= STEPS) { ++ci; s -= STEPS; }
        if (ci < p.n_conv && (C >= 32 || i < C)) {   // rows past C are the zero padding of the M tile: those lanes keep their zeros
            const unsigned* q = p.conv[ci].wp + wlane + (long)s * 512;
            hi = *reinterpret_cast<const u32x4*>(q);
            lo = *reinterpret_cast<const u32x4*>(q + 256);
        }
    };
    auto xfetch = [&](int s, int dil, u32x4 (&x1)[2], u32x4 (&x2)[2]) __attribute__((always_inline)) {
        const int kidx = 16 * s + 8 * h;
        int tap = kidx / C;
        const int ch = kidx - tap * C;
        tap = tap < K ? tap : K - 1;               // zero weight columns past the last tap: any finite row will do
        const int shift = (tap - HALF) * dil;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            int r = row0 + 32 * tt + shift;
            r = r < 0 ? 0 : (r >= p.R ? p.R - 1 : r);
            const char* a = lds + opaque_i(r * RS + ch * 2);
            x1[tt] = *reinterpret_cast<const u32x4*>(a);
            x2[tt] = *reinterpret_cast<const u32x4*>(a + plane);
        }
    };

    u32x4 wq[PD][2];
#pragma unroll
    for (int q = 0; q < PD; ++q) wq[q][0] = wq[q][1] = u32x4{0, 0, 0, 0};
    if (active) {
#pragma unroll
        for (int q = 0; q < PD; ++q) wfetch(0, q, wq[q][0], wq[q][1]);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int row = row0 + 32 * tt, pos = t0 + row;
            inside[tt] = pos >= 0 && pos < p.n;
            const float* src = p.x + ((long)b * p.n + (inside[tt] ? pos : 0)) * C + 32 * mt + 4 * h;
#pragma unroll
            for (int g = 0; g < CG; ++g) {
                xres[tt][g] = inside[tt] ? ld4(src + 8 * g) : zero4();
                put_planes(row, 32 * mt + 8 * g + 4 * h, split4(xres[tt][g]));
            }
        }
    }
    __syncthreads();

    for (int ci = 0; ci < p.n_conv; ++ci) {
        const bool last = ci + 1 == p.n_conv;
        u32x4 pl[2][CG];      // the conv's result as plane words, held across the barrier that ends everybody's reads
        if (active) {
            const int dil = p.conv[ci].dil;
            f32x16 acc[2] = {zero16(), zero16()};
            u32x4 x1[2], x2[2];
            xfetch(0, dil, x1, x2);
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const u32x4 wh = wq[s % PD][0], wl = wq[s % PD][1];
                u32x4 nh = wh, nl = wl;
                wfetch(ci, s + PD, nh, nl);                        // in flight under PD steps of MFMAs (and the epilogue)
                sched_fence();
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    acc[tt] = mfma32_f16(wh, x2[tt], acc[tt]);
                    acc[tt] = mfma32_f16(wl, x1[tt], acc[tt]);
                    acc[tt] = mfma32_f16(wh, x1[tt], acc[tt]);
                }
                wq[s % PD][0] = nh;
                wq[s % PD][1] = nl;
                if (s + 1 < STEPS) xfetch(s + 1, dil, x1, x2);
            }
            // bias, residual, zero outside the sequence; the last conv's result goes straight out
            const float* bias = p.conv[ci].bias + 32 * mt + 4 * h;
            const bool add_res = p.conv[ci].add_res != 0;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int row = row0 + 32 * tt;
                const bool emit = last && inside[tt] && row >= p.halo && row < p.halo + p.TL;
                float* o = p.out + ((long)b * p.n + (emit ? t0 + row : 0)) * C + 32 * mt + 4 * h;
#pragma unroll
                for (int g = 0; g < CG; ++g) {
                    const f32x4 bv = ld4(bias + 8 * g);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[tt][4 * g + e], kF16WScaleInv, bv[e]);
                    if (add_res) v = v + xres[tt][g];
                    if (!inside[tt]) v = zero4();
                    if (last) {
                        if (emit) {
                            if (p.accum) v = v + ld4(o + 8 * g);
                            *reinterpret_cast<f32x4*>(o + 8 * g) = v;
                        }
                    } else {
                        if (add_res) xres[tt][g] = v;
                        pl[tt][g] = split4(v);
                    }
                }
            }
        }
        if (last) break;
        __syncthreads();       // every wave has read what it needs of the old planes
        if (active) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int g = 0; g < CG; ++g) put_planes(row0 + 32 * tt, 32 * mt + 8 * g + 4 * h, pl[tt][g]);
        }
        __syncthreads();       // the new planes are complete
    }
}

// ---- C <= 16: the same kernel on v_mfma_f32_16x16x32_f16 tiles.  A 32-row MFMA tile would be 1/2 (C = 16) or 3/4 (C = 8)
// zero padding, and these two stages issue 56 % of all MFMAs of the v2 generator; 16 output channels x 16 positions x 32
// contraction elements per instruction fit C = 16 exactly and halve the matrix-pipe time of both.  A wave owns 64 positions
// = four 16-position tiles; the conv's weight fragments (<= 6 steps x 2 planes) stay in registers for the whole K loop and the
// next conv's are fetched under the epilogue and the barriers, so the K loop is LDS reads + MFMAs only.
//   A[co = lane&15][32s + 8kb + e] (kb = lane>>4),  B[32s + 8kb + e][pos = lane&15],  D: channels 4kb..4kb+3 of position lane&15
template <int C, int K>
__global__ __launch_bounds__(64 * kRbWaves, kRbWps) void hifigan_resblock16_kernel(const ResblockP p) {
    static_assert(C == 8 || C == 16, "narrow-tile kernel");
    constexpr int RS = rb_row_bytes(C), STEPS = rb_ksteps16(C, K), HALF = (K - 1) / 2;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    ESMI_DYN_LDS(lds_f);
    char* lds = reinterpret_cast<char*>(lds_f);
    const int lane = lane_id(), w = wave_id(), n = lane & 15, kb = lane >> 4;
    const bool active = w * 64 < p.R, chan_ok = 4 * kb < C;      // C = 8: k-blocks 2, 3 hold the zero rows of the M tile
    const int b = (int)blockIdx.x / p.tiles_per_b;
    const int t0 = ((int)blockIdx.x - b * p.tiles_per_b) * p.TL - p.halo;
    const int plane = p.R * RS;
    const int row0 = 64 * w + n;

    f32x4 xres[4];
    bool inside[4];
    auto split4 = [&](const f32x4& v) __attribute__((always_inline)) {
        const f32x4 a = lrelu4(v, p.slope);
        unsigned h1a, h2a, h1b, h2b;
        split_f16_pair(a[0], a[1], h1a, h2a);
        split_f16_pair(a[2], a[3], h1b, h2b);
        return u32x4{h1a, h1b, h2a, h2b};
    };
    auto put_planes = [&](int row, const u32x4& v) __attribute__((always_inline)) {
        char* d = lds + row * RS + 8 * kb;
        *reinterpret_cast<u32x2*>(d) = u32x2{v[0], v[1]};
        *reinterpret_cast<u32x2*>(d + plane) = u32x2{v[2], v[3]};
    };
    u32x4 wf[STEPS][2];
    auto wfetch = [&](int ci) __attribute__((always_inline)) {
        if (ci < p.n_conv && n < C) {              // rows past C are zero padding: those lanes keep their zeros
            const unsigned* q = p.conv[ci].wp + lane * 4;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                wf[s][0] = *reinterpret_cast<const u32x4*>(q + s * 512);
                wf[s][1] = *reinterpret_cast<const u32x4*>(q + s * 512 + 256);
            }
        }
    };
#pragma unroll
    for (int s = 0; s < STEPS; ++s) wf[s][0] = wf[s][1] = u32x4{0, 0, 0, 0};
    if (active) {
        wfetch(0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int row = row0 + 16 * nt, pos = t0 + row;
            inside[nt] = pos >= 0 && pos < p.n;
            xres[nt] = (inside[nt] && chan_ok) ? ld4(p.x + ((long)b * p.n + pos) * C + 4 * kb) : zero4();
            if (chan_ok) put_planes(row, split4(xres[nt]));
        }
    }
    __syncthreads();

    for (int ci = 0; ci < p.n_conv; ++ci) {
        const bool last = ci + 1 == p.n_conv;
        u32x4 pl[4];
        if (active) {
            const int dil = p.conv[ci].dil;
            f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const int kidx = 32 * s + 8 * kb;
                int tap = kidx / C;
                const int ch = kidx - tap * C;
                tap = tap < K ? tap : K - 1;       // zero weight columns past the last tap: any finite row will do
                const int shift = (tap - HALF) * dil;
                u32x4 x1[4], x2[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    int r = row0 + 16 * nt + shift;
                    r = r < 0 ? 0 : (r >= p.R ? p.R - 1 : r);
                    const char* a = lds + opaque_i(r * RS + ch * 2);
                    x1[nt] = *reinterpret_cast<const u32x4*>(a);
                    x2[nt] = *reinterpret_cast<const u32x4*>(a + plane);
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    acc[nt] = mfma16_f16(wf[s][0], x2[nt], acc[nt]);
                    acc[nt] = mfma16_f16(wf[s][1], x1[nt], acc[nt]);
                    acc[nt] = mfma16_f16(wf[s][0], x1[nt], acc[nt]);
                }
            }
            wfetch(ci + 1);                        // under the epilogue and the two barriers
            const f32x4 bv = chan_ok ? ld4(p.conv[ci].bias + 4 * kb) : zero4();
            const bool add_res = p.conv[ci].add_res != 0;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int row = row0 + 16 * nt;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[nt][e], kF16WScaleInv, bv[e]);
                if (add_res) v = v + xres[nt];
                if (!inside[nt]) v = zero4();
                if (last) {
                    if (chan_ok && inside[nt] && row >= p.halo && row < p.halo + p.TL) {
                        float* o = p.out + ((long)b * p.n + t0 + row) * C + 4 * kb;
                        if (p.accum) v = v + ld4(o);
                        *reinterpret_cast<f32x4*>(o) = v;
                    }
                } else {
                    if (add_res) xres[nt] = v;
                    pl[nt] = split4(v);
                }
            }
        }
        if (last) break;
        __syncthreads();       // every wave has read what it needs of the old planes
        if (active && chan_ok) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) put_planes(row0 + 16 * nt, pl[nt]);
        }
        __syncthreads();       // the new planes are complete
    }
}

}  // namespace esmi
