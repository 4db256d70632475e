// convgemm -- channels-last Conv1d / ConvTranspose1d / Linear as an implicit GEMM (fp32-accurate split products on the f16
// matrix pipe in the default build, v_mfma_f32_32x32x2_f32 in the exact-fp32 build), with the encoder-side epilogues fused:
//   out = mask( post_relu( LN( act(acc + bias) + residual ) ) ),  optional row-dot side output.
//
// One WAVE owns a 32-position x (32*NT)-channel output tile; waves are independent (no LDS, no
// barrier), a 256-thread workgroup is just four consecutive wave tiles.  Operands stream straight
// from global/L2 as 16-byte pieces: lane (i = lane&31, h = lane>>5) loads channels
// [8*kc + 4*h, +4) of input row i (A) and of weight row n (B); the four values feed four
// consecutive MFMA k-steps.  The contraction order over channels is therefore a fixed permutation
// of 0..Cin-1 -- irrelevant to the result beyond fp32 rounding.
//
// Reference ops covered (all in layers/networks.py / layers/blocks.py):
//   nn.Embedding + Conv1d (networks.py:54,64-67), qkv / proj Linear (blocks.py:44,65),
//   MixFFN Linear-Conv1d-GELU-Linear (blocks.py:22-29), LN + residual + masked_fill
//   (networks.py:73-83), Fuse Linear / ConvTranspose1d (networks.py:196-214),
//   AcousticDecoder Conv1d+ReLU+LN (networks.py:151-160).
#pragma once
#include "esmi_dev.h"

namespace esmi {

enum ConvMode { MODE_CONV = 0, MODE_CONVT = 1 };

struct ConvGemmP {
    // input rows: A + (b*n_in + t)*lda + a_coff, or embedding rows table + ids[b*n_in + t]*ld_table
    const float* A;
    int lda, a_coff;
    const int* ids;
    const float* table;
    int ld_table, vocab;
    int B, n_in, c_in, n_out, c_out;
    int k, stride, pad, mode;
    const float* W;  // (k, c_out, c_in) tap-major
    const float* bias;
    int act;
    const float* res;  // residual rows res + (b*n_out + t)*ldr + r_coff, added after act
    int ldr, r_coff;
    const float* ln_g;  // LayerNorm over c_out (requires c_out == 32*NT)
    const float* ln_b;
    int post_relu;
    const unsigned char* rowmask;  // (B, n_out), 1 => row zeroed
    float* out;                    // may be NULL (side output only)
    int ldo, o_coff;
    const float* dot_w;  // optional: dot_out[b*n_out+t] = (relu)(sum_c v[c]*dot_w[c] + dot_b[0]) on the pre-LN value
    const float* dot_b;
    float* dot_out;
    int dot_relu;
    // HiFi-GAN generator (hifigan/models.py): dilated taps, pre-activation on the INPUT rows, accumulation into `out`
    int dil;             // tap spacing (MODE_CONV): input row = t*stride + j*dil - pad; 0 is read as 1
    int act_in;          // 1: A values pass through leaky_relu(a_scale * x, act_in_slope) as they are loaded
    float act_in_slope;
    float a_scale;       // 0 is read as 1
    int accum;           // 1: out += value (same element, same thread: no race)
    // training step (data-gradient GEMMs): device pointer to max|A| (absmax_kernel).  The kernels derive the power of two s that
    // puts it in [2^9, 2^10), multiply the A values by s as they are loaded and the accumulated result by 1/s (both exact) --
    // keeps tiny gradients inside the binary16 range of the split products.  NULL: no scaling
    const float* io_scale;
    // training with `precision=16` (the reference's default, utils/tools.py:326-327 -> torch.autocast): 1 = both operands are rounded
    // to binary16 (nearest) and the contraction is ONE v_mfma_f32_32x32x16_f16 per 16 channels instead of the three products of
    // the fp32-accurate split; fp32 accumulation, fp32 in / out (master weights and activations stay fp32 in memory)
    int amp;
};
// (s, 1/s) for a tensor whose largest magnitude has the bit pattern *absmax
__device__ __forceinline__ void conv_pow2_scales(const float* absmax, float* s, float* inv) {
    const int bits = __builtin_bit_cast(int, absmax[0]);
    int e = ((bits >> 23) & 255) - 127;                  // floor(log2(absmax)) for normal numbers
    if (bits == 0 || ((bits >> 23) & 255) == 255) e = 9; // all zero / inf / nan: no scaling
    int k = 9 - e;
    k = k < -100 ? -100 : (k > 100 ? 100 : k);
    *s = __builtin_bit_cast(float, (127 + k) << 23);
    *inv = __builtin_bit_cast(float, (127 - k) << 23);
}
__device__ __forceinline__ float conv_in_scale(const ConvGemmP& p) {
    float s = 1.0f, inv = 1.0f;
    if (p.io_scale) conv_pow2_scales(p.io_scale, &s, &inv);
    return s;
}
__device__ __forceinline__ float conv_out_scale(const ConvGemmP& p) {
    float s = 1.0f, inv = 1.0f;
    if (p.io_scale) conv_pow2_scales(p.io_scale, &s, &inv);
    return inv;
}

// output positions per tile row step, and 32-row tiles per phase and batch item (see convgemm_kernel)
__host__ __device__ inline int convgemm_row_stride(const ConvGemmP& p) { return (p.mode == MODE_CONVT && p.stride > 1) ? p.stride : 1; }
__host__ __device__ inline int convgemm_tiles_per_phase(const ConvGemmP& p) {
    const int ts = convgemm_row_stride(p);
    return ((p.n_out + ts - 1) / ts + 31) >> 5;
}

// input-side activation of the HiFi-GAN convolutions, applied to a loaded fragment
__device__ __forceinline__ f32x4 conv_act_in(f32x4 v, const ConvGemmP& p, float in_s = 1.0f) {
    if (p.io_scale) v = v * in_s;
    if (p.act_in) {
        const float sc = p.a_scale != 0.0f ? p.a_scale : 1.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = v[e] * sc;
            v[e] = x > 0.0f ? x : x * p.act_in_slope;
        }
    }
    return v;
}

// Fused epilogue of the implicit-GEMM kernels, in the MFMA C/D layout (row = tile_row(r), col = n0 + 32*nt + (lane&31)):
//   out = mask( post_relu( LN( act(acc * s + bias) + residual ) ) ), optional row-dot side output on the pre-LN value.
template <int NT>
__device__ __forceinline__ void convgemm_epilogue(f32x16 (&acc)[NT], const ConvGemmP& p, int b, int t0, int n0, int lane, int ts = 1) {
    const int i = lane & 31;
    const float out_s = (ESMI_CHAIN_SPLIT ? kF16WScaleInv : 1.0f) * conv_out_scale(p);
    int col[NT];
    bool cok[NT];
    float bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        col[nt] = n0 + 32 * nt + i;
        cok[nt] = col[nt] < p.c_out;
        bias[nt] = (p.bias && cok[nt]) ? p.bias[col[nt]] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = t0 + tile_row(r, lane) * ts;
        const bool rok = t < p.n_out;
        const long row = (long)b * p.n_out + t;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float v = apply_act(fmaf(acc[nt][r], out_s, bias[nt]), p.act);
            if (p.res && rok && cok[nt]) v += p.res[row * p.ldr + p.r_coff + col[nt]];
            acc[nt][r] = v;
        }
    }
    if (p.dot_out) {
        float dw[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) dw[nt] = cok[nt] ? p.dot_w[col[nt]] : 0.0f;
        const float db = p.dot_b[0];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = 0.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) s = fmaf(acc[nt][r], dw[nt], s);
            s = row_sum32(s) + db;
            if (p.dot_relu) s = fmaxf(s, 0.0f);
            const int t = t0 + tile_row(r, lane) * ts;
            if (i == 0 && t < p.n_out) p.dot_out[(long)b * p.n_out + t] = s;
        }
    }
    if (!p.out) return;
    if (p.ln_g) layernorm_tile<NT>(acc, p.ln_g, p.ln_b, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = t0 + tile_row(r, lane) * ts;
        if (t >= p.n_out) continue;
        const long row = (long)b * p.n_out + t;
        const bool masked = p.rowmask && p.rowmask[row];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float v = acc[nt][r];
            if (p.post_relu) v = fmaxf(v, 0.0f);
            if (masked) v = 0.0f;
            if (cok[nt]) {
                float* o = p.out + row * p.ldo + p.o_coff + col[nt];
                *o = p.accum ? *o + v : v;
            }
        }
    }
}

template <int NT, bool AMP = false>     // AMP: ConvGemmP::amp as a compile-time constant (a run-time branch in the K loop broke its software pipeline: base ES +35 %)
__global__ __launch_bounds__(256) void convgemm_kernel(const ConvGemmP p) {
    const int lane = lane_id();
    // A strided ConvTranspose1d tile holds 32 positions of ONE phase (t mod stride): only the k/stride taps of that phase are
    // visited, instead of all k with (stride-1)/stride of the rows masked out of each (HiFi-GAN ups: k = 16, stride 8).
    const int ts = convgemm_row_stride(p);
    const int tiles_per_phase = convgemm_tiles_per_phase(p);
    const int tiles_per_b = tiles_per_phase * ts;
    const int wt = (int)blockIdx.x * 4 + wave_id();
    if (wt >= p.B * tiles_per_b) return;  // whole wave leaves together; no barriers in this kernel
    const int b = wt / tiles_per_b;
    const int rem = wt - b * tiles_per_b, phase = rem / tiles_per_phase;
    const int t0 = ((rem - phase * tiles_per_phase) << 5) * ts + phase;
    const int n0 = (int)blockIdx.y * (NT * 32);
    const int i = lane & 31, h = lane >> 5;
    const int t_out = t0 + i * ts;
    const int j_first = ts > 1 ? (phase + p.pad) % ts : 0;
    const float in_s = conv_in_scale(p);

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = zero16();

    const int kcs = p.c_in >> 3;
    for (int j = j_first; j < p.k; j += ts) {
        // which input row feeds output position t_out through tap j
        int ti;
        bool ok = t_out < p.n_out;
        if (p.mode == MODE_CONV) {
            ti = t_out * p.stride + j * (p.dil > 0 ? p.dil : 1) - p.pad;
            ok = ok && ti >= 0 && ti < p.n_in;
        } else {  // ConvTranspose1d: out[n*stride + j - pad] += in[n] * W[:, :, j]
            const int q = t_out + p.pad - j;
            ti = q / p.stride;
            ok = ok && q >= 0 && (q - ti * p.stride) == 0 && ti < p.n_in;
        }
        const float* arow = p.A;
        if (ok) {
            if (p.ids) {
                int id = p.ids[b * p.n_in + ti];
                if (id < 0 || id >= p.vocab) id = 0;  // the reference raises IndexError; stay in bounds
                arow = p.table + (long)id * p.ld_table;
            } else {
                arow = p.A + ((long)b * p.n_in + ti) * p.lda + p.a_coff;
            }
        }
        const float* wj = p.W + (long)j * p.c_out * p.c_in;
        const float* wrow[NT];
        bool wok[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + 32 * nt + i;
            wok[nt] = n < p.c_out;
            wrow[nt] = wj + (long)(wok[nt] ? n : 0) * p.c_in;
        }
        // KG k-steps of operands are fetched together so that one L2/HBM round trip feeds 4*KG*NT MFMAs
        // (a plain per-k-step loop serialises one memory latency per 8 channels: 4x slower at K = 32).
        constexpr int KG = NT >= 8 ? 2 : 4;
        int kc = 0;
#if ESMI_CHAIN_SPLIT
        // split-f16x2 contraction (esmi_dev.h): two k-steps (16 channels) = three v_mfma_f32_32x32x16_f16 instead of eight
        // v_mfma_f32_32x32x2_f32.  Both operands are split on the fly (the weights after the 2^8 scale; the epilogue takes it
        // out again): this plan takes the weights exactly as stored, there is no pre-split copy.  A lane's 8 k-slots of a step are
        // the channels 8kc + 4h + (0..3) and 8(kc+1) + 4h + (0..3) on BOTH sides, i.e. a fixed permutation of the 16 channels.
        auto step16 = [&](const f32x4& a0, const f32x4& a1, const f32x4 (&b0)[NT], const f32x4 (&b1)[NT]) __attribute__((always_inline)) {
            if constexpr (AMP) {   // binary16 operands, one product
                const u32x4 ah = round_f16x8(a0, a1);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32_f16(ah, round_f16x8(b0[nt] * kF16WScale, b1[nt] * kF16WScale), acc[nt]);
            } else {
                const f16x2p a2 = split_f16x2(a0, a1);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f16x2p w2 = split_f16x2(b0[nt] * kF16WScale, b1[nt] * kF16WScale);
                    acc[nt] = mfma32_split2(a2, w2.h1, w2.h2, acc[nt]);
                }
            }
        };
        for (; kc + KG <= kcs; kc += KG) {
            f32x4 av[KG], bv[KG][NT];
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                const int c = 8 * (kc + g) + 4 * h;
                av[g] = ok ? conv_act_in(ld4(arow + c), p, in_s) : zero4();
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[g][nt] = wok[nt] ? ld4(wrow[nt] + c) : zero4();
            }
#pragma unroll
            for (int g = 0; g < KG; g += 2) step16(av[g], av[g + 1], bv[g], bv[g + 1]);
        }
        for (; kc < kcs; kc += 2) {   // 8 or 16 channels left
            const int c = 8 * kc + 4 * h;
            const bool two = kc + 1 < kcs;
            const f32x4 a0 = ok ? conv_act_in(ld4(arow + c), p, in_s) : zero4(), a1 = (ok && two) ? conv_act_in(ld4(arow + c + 8), p, in_s) : zero4();
            f32x4 b0[NT], b1[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                b0[nt] = wok[nt] ? ld4(wrow[nt] + c) : zero4();
                b1[nt] = (wok[nt] && two) ? ld4(wrow[nt] + c + 8) : zero4();
            }
            step16(a0, a1, b0, b1);
        }
#else
        for (; kc + KG <= kcs; kc += KG) {
            f32x4 av[KG], bv[KG][NT];
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                const int c = 8 * (kc + g) + 4 * h;
                av[g] = ok ? conv_act_in(ld4(arow + c), p, in_s) : zero4();
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[g][nt] = wok[nt] ? ld4(wrow[nt] + c) : zero4();
            }
#pragma unroll
            for (int g = 0; g < KG; ++g) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32(av[g][s], bv[g][nt][s], acc[nt]);
                }
            }
        }
        for (; kc < kcs; ++kc) {
            const int c = 8 * kc + 4 * h;
            const f32x4 av = ok ? conv_act_in(ld4(arow + c), p, in_s) : zero4();
            f32x4 bv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = wok[nt] ? ld4(wrow[nt] + c) : zero4();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32(av[s], bv[nt][s], acc[nt]);
            }
        }
#endif
    }

    convgemm_epilogue<NT>(acc, p, b, t0, n0, lane, ts);
}


// ---- a convolution down to ONE output channel (HiFi-GAN conv_post, hifigan/models.py:123-125: 8..32 channels -> 1, k = 7,
// tanh): 1/32 of an MFMA tile's columns would be used, and the op is a plain read of the input (C floats per sample).
// One thread per output position, fp32 FMAs in tap-major / channel order; neighbouring threads share their rows in L1.
static __global__ __launch_bounds__(256) void conv_to1_kernel(const ConvGemmP p) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= (long)p.B * p.n_out) return;
    const int b = (int)(q / p.n_out), t = (int)(q - (long)b * p.n_out);
    const int dil = p.dil > 0 ? p.dil : 1;
    const float in_s = conv_in_scale(p);
    float acc = 0.0f;
    for (int j = 0; j < p.k; ++j) {
        const int ti = t + j * dil - p.pad;
        if (ti < 0 || ti >= p.n_in) continue;
        const float* arow = p.A + ((long)b * p.n_in + ti) * p.lda + p.a_coff;
        const float* wj = p.W + (long)j * p.c_in;
        for (int c = 0; c < p.c_in; c += 4) {
            const f32x4 a = conv_act_in(ld4(arow + c), p, in_s), w = ld4(wj + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = fmaf(a[e], w[e], acc);
        }
    }
    float v = apply_act(acc * conv_out_scale(p) + (p.bias ? p.bias[0] : 0.0f), p.act);
    if (p.post_relu) v = fmaxf(v, 0.0f);
    float* o = p.out + ((long)b * p.n_out + t) * p.ldo + p.o_coff;
    *o = p.accum ? *o + v : v;
}

#if ESMI_CHAIN_SPLIT
// ---- the same implicit GEMM with the WEIGHT tile staged through LDS (large shapes of the per-op plan: base ES block 1,
// long sequences).  convgemm_kernel streams both operands from L2 per wave: at 65536 rows x 3072 columns (base qkv) every
// wave re-reads its 128 weight rows for each of its row tiles, and the kernel sits at ~8 % of the matrix pipe behind L2
// latency.  Here a 256-thread workgroup owns 128 positions x BN = 32*NT channels: each wave keeps its own 32 rows (so the
// LayerNorm / row-dot epilogues stay in-wave) and reads its A fragments straight from global memory one chunk ahead, while
// the BN x 32-channel weight chunk is fetched ONCE per workgroup, split into the two f16 planes of 2^8 W once (not once per
// wave), and double-buffered in LDS: [buffer][plane][BN rows][16 data + 4 pad dwords] -- the 80-byte row stride makes both the
// ds_write_b64 of the staging threads and the 16-byte B-fragment reads (lane = weight row) bank-conflict free.
// Restrictions (the launcher falls back to convgemm_kernel otherwise): MODE_CONV, stride 1, no embedding gather, Cin % 32 == 0.
constexpr int kGemmRowDw = 20;   // dwords per weight row and plane in LDS
template <int NT>
__host__ __device__ constexpr int convgemm_lds_bytes() { return 2 * 2 * 32 * NT * kGemmRowDw * 4; }

#ifndef ESMI_GEMM_LDS_WAVES
#define ESMI_GEMM_LDS_WAVES 4   // waves (32 positions each) sharing one weight tile.  8 halves each wave's share of the staging work
                                // but couples 8 waves to one barrier: measured 11.50 vs 10.45 ms/step on base ES (r02), so 4
#endif
template <int NT, int NWV = ESMI_GEMM_LDS_WAVES, bool AMP = false>
__global__ __launch_bounds__(64 * NWV, 2) void convgemm_lds_kernel(const ConvGemmP p) {
    constexpr int BN = 32 * NT, PLANE = BN * kGemmRowDw, NTHR = 64 * NWV, NU = (256 * NT) / NTHR, ROWS = 32 * NWV;
    static_assert(NU * NTHR == 256 * NT, "staging items divide evenly");
    ESMI_DYN_LDS(lds);
    unsigned* wt = reinterpret_cast<unsigned*>(lds);   // [2 buffers][2 planes][PLANE]
    const int tid = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const int i = lane & 31, h = lane >> 5;
    const int tiles_per_b = (p.n_out + ROWS - 1) / ROWS;
    const int b = (int)blockIdx.x / tiles_per_b;
    const int t0 = ((int)blockIdx.x - b * tiles_per_b) * ROWS + 32 * w;   // this wave's 32 positions
    const int n0 = (int)blockIdx.y * BN;
    const int t_out = t0 + i;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = zero16();
    const int kchunks = p.c_in >> 5, n_it = p.k * kchunks;
    f32x4 a_nxt[2][2], w_nxt[NU];
    f16x2p a_cur[2];
    const float in_s = conv_in_scale(p);
    auto fetch = [&](int it) __attribute__((always_inline)) {   // global -> registers: A rows of this wave, W rows of the workgroup
        const int j = it / kchunks, c = (it - j * kchunks) << 5;
        const int ti = t_out + j * (p.dil > 0 ? p.dil : 1) - p.pad;
        const bool ok = t_out < p.n_out && ti >= 0 && ti < p.n_in;
        const float* arow = p.A + ((long)b * p.n_in + (ok ? ti : 0)) * p.lda + p.a_coff + c + 8 * h;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            a_nxt[st][0] = ok ? conv_act_in(ld4(arow + 16 * st), p, in_s) : zero4();
            a_nxt[st][1] = ok ? conv_act_in(ld4(arow + 16 * st + 4), p, in_s) : zero4();
        }
        const float* wj = p.W + (long)j * p.c_out * p.c_in + c;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int q = tid + NTHR * u, n = n0 + (q >> 3);
            w_nxt[u] = n < p.c_out ? ld4(wj + (long)n * p.c_in + 4 * (q & 7)) : zero4();
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {   // registers -> LDS planes (weights), A fragments split in place
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int q = tid + NTHR * u;
            const f32x4 x = w_nxt[u] * kF16WScale;
            unsigned h1a, h2a, h1b, h2b;
            split_f16_pair_rn(x[0], x[1], h1a, h2a);   // weights: nearest-rounded pieces, as the pack-time splitters
            split_f16_pair_rn(x[2], x[3], h1b, h2b);
            unsigned* d = wt + (buf * 2) * PLANE + (q >> 3) * kGemmRowDw + 2 * (q & 7);
            *reinterpret_cast<u32x2*>(d) = u32x2{h1a, h1b};
            *reinterpret_cast<u32x2*>(d + PLANE) = u32x2{h2a, h2b};
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if constexpr (AMP) a_cur[st].h1 = round_f16x8(a_nxt[st][0], a_nxt[st][1]);   // (nearest-rounded single piece; h2 unused)
            else a_cur[st] = split_f16x2(a_nxt[st][0], a_nxt[st][1]);
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    for (int it = 0; it < n_it; ++it) {
        const bool more = it + 1 < n_it;
        f16x2p a_use[2];
        a_use[0].h1 = a_cur[0].h1; a_use[1].h1 = a_cur[1].h1;
        if constexpr (!AMP) { a_use[0].h2 = a_cur[0].h2; a_use[1].h2 = a_cur[1].h2; }
        if (more) fetch(it + 1);                   // in flight under this chunk's MFMAs
        sched_fence();
        const unsigned* bp = wt + ((it & 1) * 2) * PLANE + opaque_i(i * kGemmRowDw + 4 * h);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const u32x4 b1 = *reinterpret_cast<const u32x4*>(bp + 32 * nt * kGemmRowDw + 8 * st);
                if constexpr (AMP) {   // the first plane of the staged weights IS round-to-nearest binary16 of 2^8 W
                    acc[nt] = mfma32_f16(a_use[st].h1, b1, acc[nt]);
                } else {
                    const u32x4 b2 = *reinterpret_cast<const u32x4*>(bp + PLANE + 32 * nt * kGemmRowDw + 8 * st);
                    acc[nt] = mfma32_split2(a_use[st], b1, b2, acc[nt]);
                }
            }
        }
        if (more) stage((it + 1) & 1);             // the other buffer: last read one iteration ago, before the barrier below
        __syncthreads();
    }
    if (t0 < p.n_out) convgemm_epilogue<NT>(acc, p, b, t0, n0, lane);
}
#endif

}  // namespace esmi
