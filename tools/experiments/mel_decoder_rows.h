// Row-owner form of the fused mel decoder for dx2 = 128 (tiny ES), phoneme-rate head (h0) mode -- the same function as
// mel_decoder_kernel (mel_decoder.h; MelDecoder.forward, layers/networks.py:291-304, behind the length-regulator gather,
// networks.py:233-244, and before Phoneme2Mel's masked_fill, networks.py:424-427), with the work cut the other way:
//
//   * a wave owns 32 FRAMES x ALL 128 channels and keeps them in registers from the gather to the mel store.  The products
//     are computed transposed (weights = first MFMA operand), so lane (n = lane & 31, h = lane >> 5) ends a GEMM with frame n
//     and the channels 32 tt + 8 j + 4 h + (0..3) (tt = M tile, j = register quad).  The NEXT GEMM's second operand wants 8
//     k-values per lane and 16-channel step; the weight fragments are packed with the k order permuted so that those 8 values
//     ARE registers the lane already holds (step s, element e <-> channel 16 s + 8 (e >> 2) + 4 h + (e & 3)): no activation
//     ever goes through LDS, and bias + tanh + LayerNorm (+ the block-end skip LayerNorm) run on the accumulators with two
//     v_permlane32_swap per LayerNorm as the only cross-lane traffic;
//   * the depthwise k-tap conv runs along the frames = along the lanes: DPP row shifts (row_shr / row_shl inside the 16-lane
//     DPP rows, folded into v_fmac_f32_dpp) give the +-1 / +-2 neighbours; the 2 x PAD frames a 16-frame segment needs from
//     its neighbours come from a small LDS halo buffer (every segment leaves its first / last two frames there after each
//     LayerNorm); lanes that need nothing read a zero row, so the correction is branch-free;
//   * the weights of a layer (64 KB as two f16 planes in fragment order) are staged global -> LDS once per workgroup and
//     layer and streamed by all eight waves (ds_read_b128, conflict-free);
//   * window = 256 frames (8 waves), halo = PAD * n_layers per side: 240 of 256 frames kept for tiny ES (112 of 128 in the
//     tile form), one workgroup per CU at <= 256 VGPRs, two barriers per layer (weights / halo hand-over).
//
// Weights: the rows region of the decoder blob (dec_layout: rows0 ...), written by esmi_mel_decoder_pack_f32.
#pragma once
#include "mel_decoder.h"

namespace esmi {

#ifndef ESMI_ROWS_SCHED
#define ESMI_ROWS_SCHED 1           // spell the MFMA / VALU / LDS interleave of the K loop out (sched_group_barrier)
#endif
#ifndef ESMI_ROWS_DS_PER
#define ESMI_ROWS_DS_PER 3
#endif
#ifndef ESMI_ROWS_VALU_PER
#define ESMI_ROWS_VALU_PER 10
#endif
constexpr int kRowsWin = 256;       // frames per workgroup window
constexpr int kRowsThreads = 512;
constexpr int kRowsHbRow = 132;     // floats per halo row: +4 so that the rows of one store instruction fall into different banks
constexpr int kRowsSegs = 16;       // 16-frame segments per window (+ one all-zero pseudo segment)

__host__ __device__ constexpr int dec_rows_lds_floats(int kd) {
    return 16384 + (kRowsSegs + 1) * 4 * kRowsHbRow + kd * 128 + 2 * 5 * 128;
}

// A-operand fragments of a (N, 128) row-major matrix for v_mfma_f32_32x32x16_f16 with the permuted k order (see above), two
// binary16 planes of 2^8 W (round to nearest), MT = ceil(N / 32) M tiles:
//   dst[((((s*MT + tt)*2 + pl)*64 + lane)*4 + w] = {plane(W[m][cin(2w+1)]), plane(W[m][cin(2w)])},
//   m = 32 tt + (lane & 31),  cin(e) = 16 s + 8 (e >> 2) + 4 (lane >> 5) + (e & 3)          (0 for m >= N)
static __global__ void pack_rows_afrag_kernel(const float* __restrict__ src, unsigned* __restrict__ dst, int N, int MT) {
    const long n = 8L * MT * 2 * 256;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int wd = (int)(e & 3), lane = (int)((e >> 2) & 63);
        long q = e >> 8;
        const int pl = (int)(q & 1); q >>= 1;
        const int tt = (int)(q % MT), s = (int)(q / MT);
        const int m = 32 * tt + (lane & 31);
        unsigned half[2];
        for (int j = 0; j < 2; ++j) {
            const int el = 2 * wd + j, cin = 16 * s + 8 * (el >> 2) + 4 * (lane >> 5) + (el & 3);
            const float x = (m < N ? src[(long)m * 128 + cin] : 0.0f) * kF16WScale;
            const unsigned h1 = f32_to_f16_bits(x, false);
            half[j] = pl == 0 ? h1 : f32_to_f16_bits(x - f16_bits_to_f32(h1), false);
        }
        dst[e] = half[0] | (half[1] << 16);
    }
}
// the depthwise conv's bias goes through the pointwise conv: pwb'[m] = pwb[m] + sum_k W[m][k] dwb[k]   (C = 128)
static __global__ void pack_rows_bias_kernel(const float* __restrict__ w, const float* __restrict__ dwb, const float* __restrict__ pwb,
                                      float* __restrict__ dst) {
    const int m = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (m >= 128) return;
    float a = 0.0f;
    for (int k = 0; k < 128; ++k) a = fmaf(w[m * 128 + k], dwb[k], a);
    dst[m] = pwb[m] + a;
}
// the first stage's output for a padding frame (zero input row): LN(tanh(proj_b))
static __global__ void pack_rows_padrow_kernel(const float* __restrict__ b, const float* __restrict__ g, const float* __restrict__ be,
                                        float* __restrict__ dst) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float mean = 0.0f;
    for (int c = 0; c < 128; ++c) mean += tanh_f32(b[c]);
    mean *= 1.0f / 128;
    float var = 0.0f;
    for (int c = 0; c < 128; ++c) {
        const float d = tanh_f32(b[c]) - mean;
        var = fmaf(d, d, var);
    }
    const float rstd = 1.0f / sqrtf(var * (1.0f / 128) + 1e-5f);
    for (int c = 0; c < 128; ++c) dst[c] = fmaf((tanh_f32(b[c]) - mean) * rstd, g[c], be[c]);
}

// value of lane l - D (D > 0: row_shr) or l + |D| (D < 0: row_shl) inside the 16-lane DPP row; 0 from outside the row
template <int D>
__device__ __forceinline__ float row_shift_f(float v) {
#ifdef ESMI_WAVESIM
    const int l = lane_id_raw(), src = l - D;
    const bool ok = src >= 0 && src < 64 && (src >> 4) == (l >> 4);
    const float r = wavesim::shfl(v, ok ? src : l);
    return ok ? r : 0.0f;
#else
    constexpr int ctrl = D > 0 ? 0x110 + D : 0x100 - D;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true));
#endif
}

template <int KD>
__global__ __launch_bounds__(kRowsThreads, 2) void mel_decoder_rows_kernel(const MelDecP p) {
    constexpr int C = 128, PAD = KD / 2, HB = kRowsHbRow;
    constexpr float WSI = kF16WScaleInv;
    static_assert(KD == 3 || KD == 5, "depthwise kernel sizes of the ES decoders");
    ESMI_DYN_LDS(lds);
    u32x4* wl = reinterpret_cast<u32x4*>(lds);      // the layer's weight fragments (64 KB)
    float* hb = lds + 16384;                        // halo rows [17 segments][first0, first1, last14, last15][HB]; segment 16 = zeros
    float* tapl = hb + (kRowsSegs + 1) * 4 * HB;    // [KD][128] taps of the layer whose K loop runs next
    float* epl = tapl + KD * C;                     // [2][pwb' | ln_g | ln_b | skip_g | skip_b][128], by layer parity

    const int tid = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const int n = lane & 31, h = lane >> 5, n16 = lane & 15, seg = 2 * w + ((lane >> 4) & 1);
    int tile, b;
    {   // XCD-aware workgroup -> (utterance, window) map, as in mel_decoder_kernel
        const int id = (int)blockIdx.x, per8 = 8 * p.n_tiles;
        const int g = id / per8, r = id - g * per8;
        tile = r >> 3;
        b = 8 * g + (r & 7);
        if (b >= p.B) return;
    }
    const int L = p.lmax_dev ? *p.lmax_dev : (p.lmax_host >= 0 ? p.lmax_host : batch_max_len(p.mel_len, p.B));
    const int mlen = p.mel_len ? min(p.mel_len[b], L) : L;
    const int f_lo = tile * p.TL, f0 = f_lo - p.halo;
    const int out_hi = min(f_lo + p.TL, p.L_out);
    const int valid_end = p.apply_mask ? mlen : L;
    if (f_lo >= p.L_out) return;
    if (f_lo >= valid_end) {   // whole window is padding: the final masked_fill (or the [L, L_out) tail) zeroes it
        const int cnt = (out_hi - f_lo) * p.n_mel;
        float* o = p.mel + ((long)b * p.L_out + f_lo) * p.n_mel;
        for (int e = tid; e < cnt; e += kRowsThreads) o[e] = 0.0f;
        return;
    }
    const int n_layers = p.n_blocks * p.block_depth;
    const f32x4* blob4 = reinterpret_cast<const f32x4*>(p.blob);
    const int f = f0 + 32 * w + n;                        // this lane's frame
    const bool inside = f >= 0 && f < L;
    const bool edge_window = f0 < 0 || f0 + kRowsWin > L;   // workgroup-uniform

    // ---- staging of the next layer: weight fragments by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction, no
    // staging registers, no ds_write pass; the __syncthreads that ends the phase drains them) and one float4 of the small
    // parameters per thread through a register.
    f32x4 pst = zero4();
    auto stage_fetch = [&](int l) __attribute__((always_inline)) {   // l == n_layers: the mel Linear
        const bool mel = l >= n_layers;
        const long base = mel ? p.lay.rows_mel : p.lay.rows0 + (long)l * p.lay.rows_layer_stride;
        const u32x4* g = reinterpret_cast<const u32x4*>(p.blob + base);
        const int chunks = mel ? 48 : 64;      // of 64 x 16 B
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = w + 8 * u;            // wave-uniform
            if (c < chunks) {
#ifdef ESMI_WAVESIM
                wl[c * 64 + lane] = g[c * 64 + lane];
#else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + c * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(wl + c * 64), 16, 0, 0);
#endif
            }
        }
        if (mel) {
            if (tid < C / 4) pst = blob4[(p.lay.mel_b >> 2) + tid];
        } else if (tid < (KD + 3) * (C / 4)) {
            pst = blob4[((base + 16384) >> 2) + tid];
        } else if (((l + 1) % p.block_depth) == 0 && tid < (KD + 5) * (C / 4)) {
            pst = blob4[((p.lay.skip0 + (long)(l / p.block_depth) * 2 * C) >> 2) + tid - (KD + 3) * (C / 4)];
        }
    };
    auto stage_commit = [&](int l) __attribute__((always_inline)) {
        const bool mel = l >= n_layers;
        f32x4* e4 = reinterpret_cast<f32x4*>(epl + (l & 1) * 5 * C);
        if (mel) {
            if (tid < C / 4) e4[tid] = pst;
        } else if (tid < KD * (C / 4)) {
            reinterpret_cast<f32x4*>(tapl)[tid] = pst;
        } else if (tid < (KD + 3) * (C / 4) || (((l + 1) % p.block_depth) == 0 && tid < (KD + 5) * (C / 4))) {
            e4[tid - KD * (C / 4)] = pst;
        }
    };

    // ---- halo plumbing.  Per lane: up to two rows of the neighbouring segments (A, B) and the taps that multiply them.
    int rowA = kRowsSegs * 4, rowB = kRowsSegs * 4, jA = 0, jB = 0;
    {
        const int prev = seg > 0 ? seg - 1 : kRowsSegs, next = seg < kRowsSegs - 1 ? seg + 1 : kRowsSegs;
        if (PAD == 2) {
            if (n16 == 0) { rowA = prev * 4 + 2; jA = 0; rowB = prev * 4 + 3; jB = 1; }
            else if (n16 == 1) { rowA = prev * 4 + 3; jA = 0; }
            else if (n16 == 14) { rowA = next * 4; jA = 4; }
            else if (n16 == 15) { rowA = next * 4; jA = 3; rowB = next * 4 + 1; jB = 4; }
        } else {
            if (n16 == 0) { rowA = prev * 4 + 3; jA = 0; }
            else if (n16 == 15) { rowA = next * 4; jA = 2; }
        }
    }
    const float* hpA = hb + opaque_i(rowA * HB + 4 * h);
    const float* hpB = hb + opaque_i(rowB * HB + 4 * h);
    const float* tpA = tapl + opaque_i(jA * C + 4 * h);
    const float* tpB = tapl + opaque_i(jB * C + 4 * h);
    const float* tp = tapl + opaque_i(4 * h);
    const bool halo_lane = n16 < 2 || n16 >= 14;
    float* hw = hb + opaque_i((seg * 4 + (n16 < 2 ? n16 : n16 - 12)) * HB + 4 * h);

    f32x16 xr[4], sk[4], acc[4];
    auto write_halo = [&]() __attribute__((always_inline)) {
        if (halo_lane) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = xr[tt][4 * j + e];
                    *reinterpret_cast<f32x4*>(hw + 32 * tt + 8 * j) = v;
                }
            }
        }
    };
    // LayerNorm over the 128 channels of the lane's frame: 64 here, 64 in lane ^ 32 (two-pass)
    auto ln = [&](const float* g, const float* be) __attribute__((always_inline)) {
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0 += xr[0][r]; s1 += xr[1][r]; s2 += xr[2][r]; s3 += xr[3][r];
        }
        float s = (s0 + s1) + (s2 + s3);
        s += swap32_f(s);
        const float mean = s * (1.0f / C);
        float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f, q3 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            xr[0][r] -= mean; q0 = fmaf(xr[0][r], xr[0][r], q0);
            xr[1][r] -= mean; q1 = fmaf(xr[1][r], xr[1][r], q1);
            xr[2][r] -= mean; q2 = fmaf(xr[2][r], xr[2][r], q2);
            xr[3][r] -= mean; q3 = fmaf(xr[3][r], xr[3][r], q3);
        }
        float q = (q0 + q1) + (q2 + q3);
        q += swap32_f(q);
        const float rstd = ESMI_DEC_RSQRT(q * (1.0f / C) + 1e-5f);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 32 * tt + 8 * j);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(be + 32 * tt + 8 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) xr[tt][4 * j + e] = fmaf(xr[tt][4 * j + e] * rstd, gg[e], bb[e]);
            }
        }
    };

    // ---- contraction over the 128 channels the lane pair holds in xr: DW = depthwise taps first (conv layers), MT M tiles.
    // prep(s): the second MFMA operand of 16-channel step s (depthwise conv along the lanes + split into the two f16 planes).
    auto prep = [&](auto dw_c, int s, f16x2p& a2) __attribute__((always_inline)) {
        constexpr bool DW = decltype(dw_c)::value;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int co = 16 * s + 8 * q;
            float y[4];
            if (DW) {
                f32x4 t[KD];
#pragma unroll
                for (int j = 0; j < KD; ++j) t[j] = *reinterpret_cast<const f32x4*>(tp + j * C + co);
                const f32x4 hA = *reinterpret_cast<const f32x4*>(hpA + co), tA = *reinterpret_cast<const f32x4*>(tpA + co);
                f32x4 hB = zero4(), tB = zero4();
                if (PAD == 2) {
                    hB = *reinterpret_cast<const f32x4*>(hpB + co);
                    tB = *reinterpret_cast<const f32x4*>(tpB + co);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = xr[s >> 1][8 * (s & 1) + 4 * q + e];
                    float a = t[PAD][e] * x;
                    a = fmaf(row_shift_f<1>(x), t[PAD - 1][e], a);
                    a = fmaf(row_shift_f<-1>(x), t[PAD + 1][e], a);
                    if (PAD == 2) {
                        a = fmaf(row_shift_f<2>(x), t[0][e], a);
                        a = fmaf(row_shift_f<-2>(x), t[KD - 1][e], a);
                        a = fmaf(hB[e], tB[e], a);
                    }
                    y[e] = fmaf(hA[e], tA[e], a);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = xr[s >> 1][8 * (s & 1) + 4 * q + e];
            }
            unsigned h1a, h2a, h1b, h2b;
            split_f16_pair(y[0], y[1], h1a, h2a);
            split_f16_pair(y[2], y[3], h1b, h2b);
            a2.h1[2 * q] = h1a; a2.h1[2 * q + 1] = h1b;
            a2.h2[2 * q] = h2a; a2.h2[2 * q + 1] = h2b;
        }
    };
    // Software pipeline: while step s's 3 MT MFMAs run, the VALU / LDS work of prep(s + 1) and the weight fragments of the next
    // tile are issued between them (left to itself hipcc emits `VALU block; ds_read; s_waitcnt; mfma x3; ds_read; ...` and neither
    // pipe overlaps the other inside the wave); the sched_group_barrier sequence spells the interleave out.
    auto kloop = [&](auto dw_c, auto mt_c) __attribute__((always_inline)) {
        constexpr int MT = decltype(mt_c)::value;
#pragma unroll
        for (int tt = 0; tt < MT; ++tt) acc[tt] = zero16();
        f16x2p ab[2];
        u32x4 wr[2][2];
        auto wload = [&](int idx, int slot) __attribute__((always_inline)) {
            wr[slot][0] = wl[(idx * 2 + 0) * 64 + lane];
            wr[slot][1] = wl[(idx * 2 + 1) * 64 + lane];
        };
        wload(0, 0);
        prep(dw_c, 0, ab[0]);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s + 1 < 8) prep(dw_c, s + 1, ab[(s + 1) & 1]);
#pragma unroll
            for (int tt = 0; tt < MT; ++tt) {
                const int idx = s * MT + tt;
                if (idx + 1 < 8 * MT) wload(idx + 1, (idx + 1) & 1);
                acc[tt] = mfma32_split2_wx(wr[idx & 1][0], wr[idx & 1][1], ab[s & 1], acc[tt]);
            }
#if !defined(ESMI_WAVESIM) && ESMI_ROWS_SCHED
#pragma unroll
            for (int g = 0; g < 3 * MT; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, ESMI_ROWS_DS_PER, 0);   // LDS reads
                __builtin_amdgcn_sched_group_barrier(0x002, ESMI_ROWS_VALU_PER, 0); // VALU
            }
#endif
        }
    };
    typedef std::true_type TrueC;
    typedef std::false_type FalseC;

    // ---- prologue: gather the phoneme-rate head rows (LN(tanh(proj x)) per phoneme, enc_fuse_va_kernel), stage layer 0
    stage_fetch(0);
    for (int e = tid; e < 4 * HB; e += kRowsThreads) hb[kRowsSegs * 4 * HB + e] = 0.0f;
    {
        const float* row = nullptr;
        if (inside) {
            if (f < mlen) {
                const int ph = frame_to_phoneme(p.cum + b * p.T, p.T, f);
                row = ph < p.T ? p.h0 + ((long)b * p.T + ph) * C : p.blob + p.lay.rows_pad;
            } else {
                row = p.blob + p.lay.rows_pad;   // padding frame: zero input row
            }
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = row ? ld4(row + 32 * tt + 8 * j + 4 * h) : zero4();
#pragma unroll
                for (int e = 0; e < 4; ++e) xr[tt][4 * j + e] = v[e];
            }
        }
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) sk[tt] = xr[tt];
    write_halo();
    stage_commit(0);
    __syncthreads();

    // a wave whose 32 frames all lie at or beyond L holds zeros at every stage (its halo rows were written as zeros above and
    // stay so): it only takes part in the staging and the barriers -- the last window of an utterance costs what its live waves cost
    const bool wave_live = f0 + 32 * w < L;
#ifdef ESMI_DEC_TRACE
    int tr_n = 0;
    const bool tr_on = p.trace && tile == 1 && b == p.B / 2 + 5 && lane == 0;
#define ESMI_RSTAMP() do { if (tr_on) p.trace[w * 64 + tr_n] = (long long)__builtin_amdgcn_s_memtime(); ++tr_n; } while (0)
#else
#define ESMI_RSTAMP() do {} while (0)
#endif
    for (int l = 0; l < n_layers; ++l) {
        const bool block_end = ((l + 1) % p.block_depth) == 0;
        ESMI_RSTAMP();   // 0: layer start
        if (wave_live) kloop(TrueC{}, std::integral_constant<int, 4>{});
        ESMI_RSTAMP();   // 1: K loop issued
        __syncthreads();            // every wave is done with this layer's weights, taps and halo rows
        ESMI_RSTAMP();   // 2: barrier
        stage_fetch(l + 1);
        if (!wave_live) {
            stage_commit(l + 1);
            __syncthreads();
            continue;
        }
        const float* ep = epl + opaque_i((l & 1) * 5 * C + 4 * h);
        // bias + tanh on the accumulators (the exponent's 2 log2 e is folded into the scale and the bias)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 bc = *reinterpret_cast<const f32x4*>(ep + 32 * tt + 8 * j) * kTanhExpScale;
#pragma unroll
                for (int e = 0; e < 4; ++e) xr[tt][4 * j + e] = tanh_fast_fma_f32(acc[tt][4 * j + e], WSI * kTanhExpScale, bc[e]);
            }
        }
        ESMI_RSTAMP();   // 3: fetch issued, tanh done
        ln(ep + C, ep + 2 * C);
        if (block_end) {            // skip = LN_s(x + skip), networks.py:299
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) xr[tt] += sk[tt];
            ln(ep + 3 * C, ep + 4 * C);
        }
        if (edge_window && !inside) {   // frames outside [0, L) do not exist in the reference: every Conv1d zero-pads there
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) xr[tt] = zero16();
        }
        if (block_end) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) sk[tt] = xr[tt];
        }
        ESMI_RSTAMP();   // 4: LayerNorm(s) done
        write_halo();
        stage_commit(l + 1);
        ESMI_RSTAMP();   // 5: halo + next weights written
        __syncthreads();
        ESMI_RSTAMP();   // 6: barrier
    }

    // ---- mel Linear(dx2, n_mel) on the last block's output, masked store
    if (wave_live) kloop(FalseC{}, std::integral_constant<int, 3>{});
    else {
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) acc[tt] = zero16();
    }
    if (f >= f_lo && f < out_hi) {
        const float* mb = epl + opaque_i((n_layers & 1) * 5 * C + 4 * h);
        float* orow = p.mel + ((long)b * p.L_out + f) * p.n_mel + 4 * h;
        const bool live = f < valid_end, vec_ok = (p.n_mel & 3) == 0;
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = 32 * tt + 8 * j;    // + 4 h
                if (col + 4 * h >= p.n_mel) continue;
                const f32x4 bc = *reinterpret_cast<const f32x4*>(mb + col);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = live ? fmaf(acc[tt][4 * j + e], WSI, bc[e]) : 0.0f;
                if (vec_ok) {
                    *reinterpret_cast<f32x4*>(orow + col) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + 4 * h + e < p.n_mel) orow[col + e] = v[e];
                }
            }
        }
    }
}

}  // namespace esmi
