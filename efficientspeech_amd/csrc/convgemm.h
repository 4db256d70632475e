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
    // optional: the same weights in MFMA B-fragment order with pre-split binary16 pieces (esmi_pack_bfrag_f32 of W, taps = k; the
    // `*_wp` fields of include/esmi.h).  convgemm_dma_kernel then brings its weight tile in by LDS-DMA -- each 1 KiB block of the
    // blob IS one (32-channel tile, 16-k step, piece) operand fragment in lane order -- instead of splitting fp32 rows per workgroup
    const float* Wp;
    // optional (k = 3, stride 1 convs): what the first / the last row of a sequence lacks of `bias` -- a Linear folded into the conv
    // behind it leaves its bias in every tap, and the taps that fall on the zero padding do not contribute theirs (esmi.h, ffn_cw)
    const float* bias_first;
    const float* bias_last;
    // training step, conv + LayerNorm in one launch (esmi_train_conv_ln_fwd_f32; all three NULL otherwise, all three set with ln_g):
    // what the LayerNorm's backward reads -- the normalised tensor itself (act(conv + bias) + res, (rows, c_out)), mean and rstd per row
    float* ln_pre;
    float* ln_mean;
    float* ln_rstd;
    // 1: a k = 1 problem over many rows may take pwgemm.h (whole weight resident in LDS).  Its k-slot order differs from the other
    // kernels', i.e. results differ in the last bits with the row count that selects it: set by the training step only -- the
    // inference plans promise results that do not depend on how a batch is split
    int pw_ok;
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
__device__ __forceinline__ void convgemm_epilogue(f32x16 (&acc)[NT], const ConvGemmP& p, int b, int t0, int n0, int lane, int ts = 1, int flat_rows = 0) {
    const int n_out = flat_rows ? flat_rows : p.n_out;   // (flat_rows: the caller's rows are b * n_out + t with b = 0)
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
    float b_first[NT], b_last[NT];
    const bool edge = p.bias_first != nullptr;      // kernel-uniform
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        b_first[nt] = (edge && cok[nt]) ? p.bias_first[col[nt]] : 0.0f;
        b_last[nt] = (edge && cok[nt]) ? p.bias_last[col[nt]] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = t0 + tile_row(r, lane) * ts;
        const int tl = (edge && flat_rows) ? t % p.n_out : t;   // position inside its sequence (flat_rows: t counts rows of the whole batch)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float bv = bias[nt];
            if (edge) bv = bv - (tl == 0 ? b_first[nt] : 0.0f) - (tl == p.n_out - 1 ? b_last[nt] : 0.0f);
            acc[nt][r] = apply_act(fmaf(acc[nt][r], out_s, bv), p.act);
        }
    }
    if (p.res) {   // (its own loop under a kernel-uniform branch: no load -- and no wait for one -- on the path of a launch without residual)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = t0 + tile_row(r, lane) * ts;
            if (t >= n_out) continue;
            const long row = (long)b * n_out + t;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                if (cok[nt]) acc[nt][r] += p.res[row * p.ldr + p.r_coff + col[nt]];
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
            if (i == 0 && t < n_out) p.dot_out[(long)b * n_out + t] = s;
        }
    }
    if (!p.out) return;
    // Every LOAD of this section first -- the row masks, the norm's gain and bias -- and only then stores: a load behind a store,
    // even one that a NULL pointer skips at run time, makes the compiler wait for it with vmcnt(0) behind the join, and that wait
    // also drains every store issued so far: the store loop ran one HBM write latency per row (11,000 of a decoder-size Linear's
    // 25,000 cycles in a trace).
    unsigned mbits = 0u;
    if (p.rowmask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = t0 + tile_row(r, lane) * ts;
            if (t < n_out && p.rowmask[(long)b * n_out + t]) mbits |= 1u << r;
        }
    }
    bool ln_done = false;
    if constexpr (NT <= 4) {   // (the 256-channel instantiations have no registers to spare: the training step's fused form stops at 128)
    if (p.ln_g && p.ln_pre) {
        ln_done = true;
        float gg[NT], bb[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { gg[nt] = p.ln_g[32 * nt + i]; bb[nt] = p.ln_b[32 * nt + i]; }
        // the pre-norm tensor for the backward, then the norm with its statistics written out (layernorm_tile's arithmetic)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = t0 + tile_row(r, lane) * ts;
            if (t >= n_out) continue;
            const long row = (long)b * n_out + t;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                if (cok[nt]) p.ln_pre[row * p.c_out + col[nt]] = acc[nt][r];
        }
        const float inv_c = 1.0f / (float)(32 * NT);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = 0.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) s += acc[nt][r];
            const float mean = row_sum32(s) * inv_c;
            float q = 0.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { const float d = acc[nt][r] - mean; q = fmaf(d, d, q); }
            const float rstd = 1.0f / sqrtf(row_sum32(q) * inv_c + 1e-5f);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt][r] = fmaf((acc[nt][r] - mean) * rstd, gg[nt], bb[nt]);
            const int t = t0 + tile_row(r, lane) * ts;
            if (i == 0 && t < n_out) { p.ln_mean[(long)b * n_out + t] = mean; p.ln_rstd[(long)b * n_out + t] = rstd; }
        }
    }
    }
    if (p.ln_g && !ln_done) layernorm_tile<NT>(acc, p.ln_g, p.ln_b, lane);
    if (p.accum) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = t0 + tile_row(r, lane) * ts;
            if (t >= n_out) continue;
            const long row = (long)b * n_out + t;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float v = acc[nt][r];
                if (p.post_relu) v = fmaxf(v, 0.0f);
                if ((mbits >> r) & 1u) v = 0.0f;
                if (cok[nt]) {
                    float* o = p.out + row * p.ldo + p.o_coff + col[nt];
                    *o = *o + v;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = t0 + tile_row(r, lane) * ts;
        if (t >= n_out) continue;
        const long row = (long)b * n_out + t;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float v = acc[nt][r];
            if (p.post_relu) v = fmaxf(v, 0.0f);
            if ((mbits >> r) & 1u) v = 0.0f;
            if (cok[nt]) p.out[row * p.ldo + p.o_coff + col[nt]] = v;
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
        const float* wj = p.W ? p.W + (long)j * p.c_out * p.c_in : nullptr;   // (NULL: the caller gave the pre-split blob alone -- the `pre` loop below covers all of K)
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
        // With the caller's pre-split blob (`Wp`, esmi_pack_bfrag_f32 of W: each 1 KiB block IS one (16-channel step, piece, column tile)
        // B fragment in lane order, and its k-slot order -- channels 16 st + 4 h + (0..3) and + 8 -- is this kernel's A order) the weights
        // arrive as one coalesced 16-byte load per lane, piece and tile, and nothing is split: the loop above spends ~58 VALU instructions
        // per column tile and 16 channels on the weights (three MFMAs' worth of issue time four times over at NT = 4), the same for every
        // wave of the launch.  (Round 5; the blob's pieces are nearest-rounded, the on-the-fly ones truncated: last-bit differences.)
        const bool pre = p.Wp != nullptr && (p.c_in & 31) == 0;          // (kernel-uniform)
        if (pre) {
            const int ntw = (p.c_out + 31) >> 5, tile0 = n0 >> 5;
            for (; kc + KG <= kcs; kc += KG) {
                f32x4 av[KG];
                u32x4 wb[KG / 2][NT][AMP ? 1 : 2];
#pragma unroll
                for (int g = 0; g < KG; ++g) av[g] = ok ? conv_act_in(ld4(arow + 8 * (kc + g) + 4 * h), p, in_s) : zero4();
#pragma unroll
                for (int s2 = 0; s2 < KG / 2; ++s2)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int pl = 0; pl < (AMP ? 1 : 2); ++pl) {
                            const long blk = ((long)j * (p.c_in >> 3) + kc + 2 * s2 + pl) * ntw + tile0 + nt;
                            wb[s2][nt][pl] = tile0 + nt < ntw ? __builtin_bit_cast(u32x4, ld4(p.Wp + (blk * 64 + lane) * 4)) : u32x4{0u, 0u, 0u, 0u};
                        }
#pragma unroll
                for (int s2 = 0; s2 < KG / 2; ++s2) {
                    if constexpr (AMP) {
                        const u32x4 ah = round_f16x8(av[2 * s2], av[2 * s2 + 1]);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32_f16(ah, wb[s2][nt][0], acc[nt]);
                    } else {
                        const f16x2p a2 = split_f16x2(av[2 * s2], av[2 * s2 + 1]);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32_split2(a2, wb[s2][nt][0], wb[s2][nt][AMP ? 0 : 1], acc[nt]);
                    }
                }
            }
        }
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
// ---- convgemm_dma_kernel: the same implicit GEMM with BOTH operands staged through LDS (large shapes of the per-op plan: base ES
// block 1, long sequences, the training step's GEMMs).  convgemm_kernel streams both operands from L2 per wave: at 65536 rows x
// 3072 columns (base qkv) every wave re-reads its 128 weight rows for each of its row tiles, and the kernel sits at ~8 % of the
// matrix pipe behind L2 latency.  Here a 256-thread workgroup owns 32 MT x 4 rows (the FLAT index b * n_out + t: a tile may span
// utterances, so 64-position sequences still fill it) x BN = 32 NT channels; each wave keeps its own 32 MT rows, so the LayerNorm /
// row-dot epilogues stay in-wave (NT = 8, MT = 1 when they span 256 channels; else NT = 4 and MT = 2 when the grid is large enough).
//  * weight tile, double-buffered: either split per workgroup from the fp32 rows into the two f16 planes of 2^8 W
//    ([buffer][plane][BN rows][16 data + 4 pad dwords]: the 80-byte row stride keeps the staging ds_write_b64 and the 16-byte
//    B-fragment reads conflict-free), or -- when the caller has the esmi_pack_bfrag_f32 blob (`Wp`) -- copied by LDS-DMA: each 1 KiB
//    block of the blob IS one (tile, 16-k step, piece) fragment in lane order.
//  * input rows by LDS-DMA.  Round 2's version read the A fragments from global memory into registers; ablations on it
//    (profiles/r03_probes/gemm_lds_ablation.md) put its time there: a lane of the MFMA operand layout owns 32 contiguous bytes of
//    ITS OWN row, so every global_load_dwordx4 of a wave touched 32 different cache lines for 16 bytes each (no A loads: -30 %; no
//    weight loads: -8 %; no barrier: -3 %).  Now eight lanes fetch one whole 128-byte row chunk (8 lines per instruction) with
//    global_load_lds_dwordx4 -- memory -> LDS, no staging registers -- into a wave-PRIVATE fp32 tile, and the fragments are read
//    back with ds_read_b128.  LDS-DMA writes lane l's 16 bytes at base + 16 l, so the tile is plain row-major [row][8 pieces]; the
//    bank spread comes from WHICH piece a lane fetches instead: slot s of row r holds piece s ^ ((r >> 1) & 7), and 16 consecutive
//    rows reading the same piece hit 16 different 16-byte bank groups.  The tile is single-buffered: a wave reads its fragments of
//    chunk c into registers, waits for them (lgkmcnt), and only then issues the DMA of chunk c + 1 into the same rows -- no other
//    wave touches them.  Rows whose tap falls outside the utterance (or past the last row) are fetched from a clamped in-range
//    address and zeroed by the reader, which knows its own row's position.
//  * convolution taps re-use the rows: when the taps' reach (k - 1) * dil is at most kGemmHaloMax rows the wave's tile carries that
//    many extra rows, the K loop runs channel chunk OUTER / tap INNER, and one DMA per chunk serves all k taps (tap j reads the
//    fragments j * dil rows further down) -- a third of the input traffic of a k = 3 convolution.  Otherwise (long dilated HiFi-GAN
//    taps) each tap fetches its own shifted rows, tap outer.
//  * workgroups are numbered so that the c_out / BN column tiles of one row tile are neighbours ON THE SAME XCD (ids congruent
//    mod 8 share an XCD and its L2): they sweep the same input rows at the same time and three of four fetches hit L2.
// Restrictions (the launcher falls back to convgemm_kernel otherwise): MODE_CONV, stride 1, n_in == n_out, no embedding gather,
// Cin % 32 == 0, an input tensor of < 2^31 elements (32-bit lane offsets).
constexpr int kGemmRowDw = 20;   // dwords per weight row and plane in LDS (weights split in the kernel)
template <int NT>
__host__ __device__ constexpr int convgemm_lds_bytes() { return 2 * 2 * 32 * NT * kGemmRowDw * 4; }

constexpr int kGemmLdsWaves = 4;   // waves sharing one weight tile.  8 halves each wave's share of the staging work but couples 8 waves
                                   // to one barrier: 11.50 vs 10.45 ms/step on base ES in round 2, +2 % on the round-3 kernel: 4
#ifdef ESMI_GEMM_TRACE
extern __device__ long long* g_gemm_trace_dev;
#endif
constexpr int kGemmHaloMax = 16;
__host__ __device__ inline int convgemm_dma_tile_rows(int mt, int k, int dil) {   // LDS rows of one wave's input tile
    const int reach = (k - 1) * (dil > 0 ? dil : 1);
    return 32 * mt + (reach <= kGemmHaloMax ? (reach + 7) / 8 * 8 : 0);
}
template <int NT>
__host__ __device__ inline int convgemm_dma_bytes(int mt, int k, int dil, bool pre, int nwv = kGemmLdsWaves) {
    return (pre ? 2 * 4 * NT * 1024 : convgemm_lds_bytes<NT>()) + nwv * convgemm_dma_tile_rows(mt, k, dil) * 128;
}

template <int NT, int MT, int NWV = kGemmLdsWaves, bool AMP = false, bool PRE = false>
__global__ __launch_bounds__(64 * NWV, 2) void convgemm_dma_kernel(const ConvGemmP p, int nx, int ny) {
    constexpr int BN = 32 * NT, PLANE = BN * kGemmRowDw, NTHR = 64 * NWV, NU = (256 * NT) / NTHR, WROWS = 32 * MT, ROWS = WROWS * NWV;
    static_assert(NU * NTHR == 256 * NT, "staging items divide evenly");
    static_assert(MT <= 2, "epilogue calls below");
    // workgroup id -> (row tile bx, column tile by): id = 8 * (ny * xs + by) + xcd, bx = 8 * xs + xcd
    const int xcd = (int)blockIdx.x & 7, nq = (int)blockIdx.x >> 3;
    const int by = nq % ny, bx = (nq / ny) * 8 + xcd;
    if (bx >= nx) return;                              // (the grid is padded to whole groups of 8 row tiles)
    ESMI_DYN_LDS(lds);
    unsigned* wt = reinterpret_cast<unsigned*>(lds);   // [2 buffers][2 planes][PLANE]
    const int tid = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const int dil = p.dil > 0 ? p.dil : 1;
    const int reach = (p.k - 1) * dil;
    const bool halo = reach <= kGemmHaloMax;           // taps share one fetch per channel chunk
    const int trows = convgemm_dma_tile_rows(MT, p.k, dil), nd = trows >> 3;
    float* at = lds + (PRE ? 8 * NT * 256 : 4 * PLANE) + w * (trows * 32);    // this wave's input rows: [trows][8 pieces of 4 floats, swizzled]
    const int i = lane & 31, h = lane >> 5;
    const long n_rows = (long)p.B * p.n_out;           // == B * n_in (the launcher checks n_in == n_out): input row = output row + tap offset
    const long r0 = (long)bx * ROWS + WROWS * w;       // this wave's first flat row b * n_out + t
    const int n0 = by * BN;
    constexpr int WBLK = 256, WBUF = PRE ? 4 * NT * WBLK : 2 * PLANE;   // dwords: one fragment block / one weight buffer
    (void)WBUF;
    // reader role: MFMA operand layout, lane (i, h) <-> row 32 mt + i, channels 8 h + 16 st + (0..7)
    int rt[MT];
    bool rok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long r = r0 + 32 * mt + i;
        rok[mt] = r < n_rows;
        rt[mt] = rok[mt] ? (int)(r % p.n_out) : 0;
    }
    // loader role: DMA instruction d, lane l <-> tile row 8 d + (l >> 3), LDS slot l & 7, which holds piece (l & 7) ^ ((row >> 1) & 7)
    const int l_row = lane >> 3;
    const int piece_e = (lane & 7) ^ (lane >> 4), piece_o = piece_e ^ 4;   // d even / odd
    const int lbase = ((int)r0 + l_row) * p.lda + p.a_coff;                // (32-bit: the launcher checks the tensor's span)
    const int max_off = ((int)n_rows - 1) * p.lda + p.a_coff + p.c_in - 4;

    f32x16 acc[MT * NT];   // tile (mt, nt) at mt * NT + nt
#pragma unroll
    for (int q = 0; q < MT * NT; ++q) acc[q] = zero16();
    const int kchunks = p.c_in >> 5, n_it = p.k * kchunks;
    f32x4 w_nxt[NU];
    const float in_s = conv_in_scale(p);
    // iteration -> (tap j, channel chunk c): halo mode runs the taps innermost
    int nj = 0, nc = 0;   // the NEXT iteration's (plain loop-carried scalars: behind a by-reference lambda they ended up in scratch)
    // input rows of tap j / chunk c: memory -> LDS (DMA instruction d, lane l: tile row 8 d + (l >> 3))
    auto fetch_a = [&](int j, int c) __attribute__((always_inline)) {
        const int shift = halo ? -p.pad : j * dil - p.pad;    // first tile row, relative to the wave's first output row
        const int off0 = lbase + shift * p.lda + (c << 5);
        for (int d = 0; d < nd; ++d) {
            int off = off0 + 8 * d * p.lda + 4 * ((d & 1) ? piece_o : piece_e);
            off = off < 0 ? 0 : (off > max_off ? max_off : off);
            lds_dma16(p.A + off, at + 8 * d * 32, lane);
        }
    };
    const int ntw = (p.c_out + 31) >> 5;
    // weight tile of the workgroup for tap j / chunk c into buffer `buf`
    auto fetch_w = [&](int j, int c, int buf) __attribute__((always_inline)) {
        if constexpr (PRE) {   // blob -> LDS: blocks [16-k step st][piece][tile nt] of 64 lanes x 16 bytes, lane order
            constexpr int P = AMP ? 1 : 2, NB = 2 * P * NT;
            const long blk0 = ((long)j * (p.c_in >> 3) + 4 * c) * ntw + (n0 >> 5);
            unsigned* dstb = wt + buf * WBUF;
#pragma unroll
            for (int v0 = 0; v0 < NB; v0 += NWV) {
                const int v = v0 + w;                      // (wave-uniform)
                if (v < NB) {
                    const int st = v / (P * NT), pl = (v / NT) % P, nt = v % NT;
                    unsigned* d = dstb + ((st * 2 + pl) * NT + nt) * WBLK;
                    if ((n0 >> 5) + nt < ntw) {
                        const float* src = p.Wp + ((blk0 + (long)(2 * st + pl) * ntw + nt) * 64 + lane) * 4;
                        lds_dma16(src, d, lane);
                    } else {
                        reinterpret_cast<u32x4*>(d)[lane] = u32x4{0u, 0u, 0u, 0u};   // column tile past c_out
                    }
                }
            }
        } else {               // fp32 rows -> registers (split and written to LDS by stage())
            const float* wj = p.W + (long)j * p.c_out * p.c_in + (c << 5);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int q = tid + NTHR * u, n = n0 + (q >> 3);
                w_nxt[u] = n < p.c_out ? ld4(wj + (long)n * p.c_in + 4 * (q & 7)) : zero4();
            }
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {   // registers -> LDS planes (weights)
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        if constexpr (PRE) return;
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
    };
#ifdef ESMI_GEMM_TRACE   // development: shader-clock stamps of one mid-grid workgroup, 7 per K chunk (tools/trace_gemm.py)
    const bool tr_on = g_gemm_trace_dev && p.k == 3 && p.c_in == 512 && bx == nx / 2 && by == 0 && lane == 0;
    int tr_n = 0;
#define ESMI_GT() do { if (tr_on && tr_n < 512) g_gemm_trace_dev[w * 512 + tr_n] = (long long)__builtin_amdgcn_s_memtime(); ++tr_n; } while (0)
#else
#define ESMI_GT() do {} while (0)
#endif
    fetch_a(0, 0);
    fetch_w(0, 0, 0);
    stage(0);
    __syncthreads();
    for (int it = 0; it < n_it; ++it) {
        const int tj = nj;                             // this iteration's tap; (nj, nc) := the next iteration's
        if (halo) { const bool wrap = nj + 1 == p.k; nj = wrap ? 0 : nj + 1; nc += wrap ? 1 : 0; }
        else { const bool wrap = nc + 1 == kchunks; nc = wrap ? 0 : nc + 1; nj += wrap ? 1 : 0; }
        const bool more = it + 1 < n_it;
        const bool more_a = more && (!halo || nj == 0);   // halo mode: new rows only when the next iteration starts a chunk
        f16x2p a_use[MT][2];
        ESMI_GT();   // 0: top
        {   // this chunk's operand fragments: all LDS reads first, ONE wait, then zero the rows outside the utterance and split
            const int tap = tj * dil - p.pad;
            const int row_l = i + (halo ? tap + p.pad : 0);             // tile row of this lane's first fragment row
            const float* ap = at + row_l * 32;
            const int swz = (row_l >> 1) & 7;
            f32x4 raw[MT][2][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    // k-slots of a lane: channels 16 st + 8 h + (0..7) against weights split in this kernel, 16 st + 4 h + (0..3)
                    // and + 8 against the packed blob (the order esmi_pack_bfrag_f32 writes)
                    const int s0 = (PRE ? 4 * st + h : 4 * st + 2 * h) ^ swz;   // (32 mt rows further: same (row >> 1) & 7)
                    raw[mt][st][0] = *reinterpret_cast<const f32x4*>(ap + 32 * mt * 32 + 4 * s0);
                    raw[mt][st][1] = *reinterpret_cast<const f32x4*>(ap + 32 * mt * 32 + 4 * (s0 ^ (PRE ? 2 : 1)));
                }
            sched_fence();
            const bool plain = !p.act_in && !p.io_scale;   // (wave-uniform)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int ti = rt[mt] + tap;
                const unsigned keep = (rok[mt] && ti >= 0 && ti < p.n_in) ? 0xffffffffu : 0u;
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    f32x4 lo = raw[mt][st][0], hi = raw[mt][st][1];
                    if (!plain) { lo = conv_act_in(lo, p, in_s); hi = conv_act_in(hi, p, in_s); }
                    lo = __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, lo) & keep);
                    hi = __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, hi) & keep);
                    if constexpr (AMP) a_use[mt][st].h1 = round_f16x8(lo, hi);
                    else a_use[mt][st] = split_f16x2(lo, hi);
                }
            }
        }
        lds_wave_sync();                           // the fragment reads have returned: the rows may be overwritten
        sched_fence();
        ESMI_GT();   // 1: A fragments read + split
        if (more_a) fetch_a(nj, nc);               // the next iteration's loads: in flight under this iteration's products
        if (more) fetch_w(nj, nc, (it + 1) & 1);
        sched_fence();
        ESMI_GT();   // 2: loads issued
        const unsigned* bp = PRE ? wt + (it & 1) * WBUF + opaque_i(4 * lane) : wt + ((it & 1) * 2) * PLANE + opaque_i(i * kGemmRowDw + 4 * h);
        // B fragments in groups of two 32-channel tiles, one group ahead of the products that use them (hipcc on its own issues
        // each pair of reads right in front of its products and waits for the full LDS round trip every time)
        constexpr int NG = NT;   // 2 k-halves x NT / 2 tile pairs
        u32x4 bf[2][2][2];       // [group parity][tile of the pair][plane]
        auto ld_b = [&](int g) __attribute__((always_inline)) {
            const int st = g / (NT / 2), np = g % (NT / 2);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned* src = PRE ? bp + (st * 2 * NT + 2 * np + q) * WBLK : bp + 32 * (2 * np + q) * kGemmRowDw + 8 * st;
                bf[g & 1][q][0] = *reinterpret_cast<const u32x4*>(src);
                if constexpr (!AMP) bf[g & 1][q][1] = *reinterpret_cast<const u32x4*>(src + (PRE ? NT * WBLK : PLANE));
            }
        };
        ld_b(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) ld_b(g + 1);
            sched_fence();
            const int st = g / (NT / 2), np = g % (NT / 2);
            if constexpr (AMP) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        f32x16& c = acc[mt * NT + 2 * np + q];
                        c = mfma32_f16(a_use[mt][st].h1, bf[g & 1][q][0], c);
                    }
            } else {
                // the three products of a tile are a dependent chain on its accumulator: run the 2 MT chains side by side
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            f32x16& c = acc[mt * NT + 2 * np + q];
                            const f16x2p& a = a_use[mt][st];
                            c = pr == 0 ? mfma32_f16(a.h2, bf[g & 1][q][0], c) : pr == 1 ? mfma32_f16(a.h1, bf[g & 1][q][1], c) : mfma32_f16(a.h1, bf[g & 1][q][0], c);
                        }
            }
            sched_fence();
        }
        ESMI_GT();   // 3: products issued
#ifdef ESMI_GEMM_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ESMI_GT();   // 4: weight rows + DMA arrived
#endif
        if (more) stage((it + 1) & 1);             // the other buffer: last read one iteration ago, before the barrier below
        ESMI_GT();   // 5: weights staged
        __syncthreads();                           // (its fence also drains this wave's DMA: vmcnt(0))
        ESMI_GT();   // 6: barrier passed
    }
    if (r0 < n_rows) convgemm_epilogue<NT>(*reinterpret_cast<f32x16(*)[NT]>(&acc[0]), p, 0, (int)r0, n0, lane, 1, (int)n_rows);
    if constexpr (MT > 1) {
        if (r0 + 32 < n_rows) convgemm_epilogue<NT>(*reinterpret_cast<f32x16(*)[NT]>(&acc[NT]), p, 0, (int)r0 + 32, n0, lane, 1, (int)n_rows);
    }
}
#endif

}  // namespace esmi
