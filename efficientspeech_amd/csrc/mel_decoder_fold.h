// Fully fused mel decoder, LayerNorm-folded form -- MelDecoder.forward, layers/networks.py:291-304, plus the length
// regulator gather (networks.py:233-244) on its input side and Phoneme2Mel's final masked_fill
// (networks.py:424-427) on its output side.
//
//   skip = LN(tanh(Linear(d4,dx2)(x)))
//   n_blocks x { x = skip; block_depth x [ x = LN(tanh(Conv1x1(dwConv_k(x)))) ]; skip = LN_s(x + skip) }
//   mel  = Linear(dx2, n_mel)(skip)
//
// One 512-thread workgroup (8 waves) owns a 128-frame window of one utterance: TL = 128 - 2*halo
// frames are kept, halo = (k/2)*n_blocks*block_depth frames per side are recomputed so that no
// activation ever leaves the CU.  Activations live in ONE LDS tile [132][DX2+4] fp32 (two zero rows
// per side = the depthwise conv's in-window padding; the +4 floats/row keep the 16-byte row-fragment
// reads bank-conflict free AND hold the row's LayerNorm statistics, see below).
//
// What "folded" means.  A LayerNorm output  y[p][c] = (x[p][c] - mu_p) * r_p * g[c] + b[c]  is consumed here by exactly one
// operation: the next depthwise conv (or the mel Linear).  So the tile never holds y.  It holds the RAW rows x (tanh outputs, or
// LN(x) + skip at a block end) plus, per row, the two scalars (r_p, -mu_p r_p); the consumer normalises while it loads
// (one fma per element: n = x r_p - mu_p r_p) and the per-channel gain / shift are folded into ITS parameters at pack time:
//     dw(y)[t][c] = dw_b[c] + sum_k w[k][c] y[t+k][c]  =  c0[c] + sum_k (w[k][c] g[c]) n[t+k][c],   c0 = dw_b + b sum_k w[k]
// (rows outside [0, L) must contribute 0, not b: the few output rows next to an utterance end subtract the missing taps' share,
// `corr`, in edge windows only), and  mel = (W diag(g_s)) n + (W b_s + b_mel).  The statistics come from the accumulators right
// behind the tanh: a wave holds 32*NTW of a row's dx2 channels, reduces them in-lane + one v_permlane32_swap (two-pass: mean,
// centred squares) and leaves (mean_w, M2_w) in LDS; whoever needs the row merges the four partials (Chan).  The skip tensor
// lives in the ACCUMULATOR layout (same registers as before, no row-owner threads any more).  Compared with the row-owner
// LayerNorm pass of rounds 1-3 this removes, per layer, one full read + write of the tile, the normalise + gain + shift
// arithmetic on it, and one workgroup barrier (block ends: LN(x) + skip runs on the accumulators).
//
// Weight-stationary GEMMs (ESMI_DEC_SPLIT: exact-fp32 MFMA, or fp32-accurate split products on the f16 matrix pipe).  Wave
// w = (mh = w>>2, ns = w&3) owns rows [64mh, 64mh+64) x columns [ns*DX2/4, +DX2/4).  For each 128-channel K chunk it loads
// its weight slice into registers (coalesced 16-byte buffer loads from the pre-packed blob) and streams the A fragments of
// its 64 rows from LDS: the K loop touches no global memory.
//
// Per conv layer, four phases separated by workgroup barriers (five at a block end):
//   A1. per row group: merge the statistics of the rows it is about to read -> the rows' pad floats; load the window (12 or 20
//       rows x 4 channels per thread), normalise it                                                            | barrier
//   A2. depthwise k-tap conv from registers, written IN PLACE as the K loop's operand (already split in two f16 planes) | barrier
//   K.  K loop (MFMA + ds_read_b128 only); bias + tanh on the accumulators; partial statistics -> LDS            | barrier
//   S.  accumulators -> tile (raw rows)                                                                          | barrier
//   block end, between K and S: merge own rows | barrier | x = LN(t), u = x + skip, partial statistics of u;  S stores u;
//   behind S's barrier: skip = LN_s(u) on the accumulators.
//
// Fidelity notes (SURVEY.md §7 "hard parts"):
//  * frames in [mel_len[b], L) are PADDING FRAMES: zero input rows, but computed like any other frame,
//    because the reference computes them and the k-tap conv leaks them into the last valid frames;
//  * frames outside [0, L) do not exist in the reference: every layer's Conv1d zero-pads there, so
//    such rows get the statistics (0, 0) -- they normalise to 0 -- and their neighbours the `corr` term;
//  * rows >= mel_len[b] of the output are zeroed only at the very end (the final masked_fill).
#pragma once
#include <type_traits>

#include "esmi_dev.h"
#include "small_kernels.h"

// Build knobs (defaults = the measured best on MI355X; history in DESIGN.md 3.1)
#ifndef ESMI_DEC_SPLIT      // contraction of the pointwise GEMMs (esmi_dev.h):
#define ESMI_DEC_SPLIT 2    //   0: v_mfma_f32_32x32x2_f32 (exact fp32; the libesmi_fp32mfma.so build)
#endif                      //   2: fp32 split into 2 f16 (weights pre-scaled by 2^8), 3 products on v_mfma_f32_32x32x16_f16
#if ESMI_DEC_SPLIT != 0 && ESMI_DEC_SPLIT != 2
#error "ESMI_DEC_SPLIT must be 0 (fp32 MFMA) or 2 (split f16x2)"
#endif
#ifndef ESMI_DEC_WD256
#define ESMI_DEC_WD256 2    // weight-fragment ring depth (16-channel steps) of the dx2 = 256 kernel's K loop (one workgroup per CU)
#endif

namespace esmi {

constexpr int kDecRows = 128;     // frames per workgroup window
constexpr int kDecPadRows = 2;    // zero rows above/below the window in LDS (>= k/2)
constexpr int kDecThreads = 512;  // 8-wave windows
constexpr int kMelCols = 96;      // n_mel <= 96 (three 32-column MFMA tiles)

struct DecLayout {  // offsets in floats into the packed blob
    long proj_w, proj_b, proj_g, proj_beta;    // proj_b, proj_g, proj_beta contiguous
    long h0_pad;                               // LN(tanh(proj_b)): the first-stage row of a padding frame (zero input row)
    long layer0, layer_stride;                 // per conv layer, relative to the layer base:
    long l_taps, l_c0;                         //   group A (contiguous): folded taps [kd][dx2], c0 [dx2]
    long l_pwb, l_g, l_b, l_sg, l_sb;          //   group B (contiguous): pw_b, ln_g, ln_b, skip_g, skip_b (the skip pair: block ends only)
    long l_pw;                                 //   the packed pointwise matrix
    long layer0_h0;                            // group A of layer 0 when its input is already normalised (h0 supplied)
    long mel_w, mel_b;                         // folded with the last skip LayerNorm
    long total;
    int ga, gb;                                // floats in group A / group B
};

inline DecLayout dec_layout(int d4, int dx2, int kd, int n_blocks, int block_depth) {
    DecLayout L;
    long o = 0;
    constexpr long kWNum = 2;   // matrix storage in units of DX2*DX2/2 floats: two f16 planes or fp32 (the same bytes)
    L.proj_w = o; o += (long)d4 * dx2 * kWNum / 2;
    L.proj_b = o; o += dx2;
    L.proj_g = o; o += dx2;
    L.proj_beta = o; o += dx2;
    L.h0_pad = o; o += dx2;
    L.l_taps = 0;
    L.l_c0 = (long)kd * dx2;
    L.ga = (kd + 1) * dx2;
    L.l_pwb = L.ga;
    L.l_g = L.l_pwb + dx2;
    L.l_b = L.l_g + dx2;
    L.l_sg = L.l_b + dx2;
    L.l_sb = L.l_sg + dx2;
    L.gb = 5 * dx2;
    L.l_pw = L.l_pwb + L.gb;
    L.layer_stride = L.l_pw + (long)dx2 * dx2 * kWNum / 2;
    L.layer0 = o; o += L.layer_stride * n_blocks * block_depth;
    L.layer0_h0 = o; o += L.ga;
    L.mel_w = o; o += (long)dx2 * dx2 * kWNum / 2;   // packed like a dx2 x dx2 matrix, rows >= n_mel zero
    L.mel_b = o; o += dx2;                     // zero padded
    L.total = o;
    return L;
}

// A LayerNorm gain of exactly 0 (a dead channel) is replaced by 1e-30: the rows outside [0, L) are stored as -b/g so that the
// folded taps cancel the folded shift there, which needs g != 0; the output changes by 1e-30 * O(1).
__host__ __device__ inline float nonzero_gain(float g) { return fabsf(g) < 1e-30f ? (g < 0.0f ? -1e-30f : 1e-30f) : g; }

// Weight-stationary B-fragment packing of a (N, K) row-major matrix, K a multiple of 128, for a
// workgroup whose 4 column slices are WCOLS = 32*NTW wide (`cs`: optional per-input-channel scale, the folded LayerNorm gain):
//   dst[(((((c*4 + ns)*NTW + ntw)*16 + kc)*64 + lane)*4 + s] =
//       W[ns*WCOLS + 32*ntw + (lane&31)][128*c + 8*kc + 4*(lane>>5) + s]      (0 for rows >= N)
static __global__ void pack_bslice_kernel(const float* __restrict__ src, const float* __restrict__ cs, float* __restrict__ dst, int N, int K, int NTW) {
    const long n = (long)(K / 128) * 4 * NTW * 16 * 256;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int s = (int)(e & 3);
        const int lane = (int)((e >> 2) & 63);
        long q = e >> 8;
        const int kc = (int)(q & 15); q >>= 4;
        const int ntw = (int)(q % NTW); q /= NTW;
        const int ns = (int)(q & 3);
        const int c = (int)(q >> 2);
        const int row = ns * 32 * NTW + 32 * ntw + (lane & 31);
        const int col = 128 * c + 8 * kc + 4 * (lane >> 5) + s;
        dst[e] = row < N ? src[(long)row * K + col] * (cs ? nonzero_gain(cs[col]) : 1.0f) : 0.0f;
    }
}

// The same slices as two binary16 planes of 2^8 * W (round to nearest; esmi_dev.h) in the B layout of v_mfma_f32_32x32x16_f16:
// per (chunk c, column slice ns, tile ntw, 16-channel step s, plane p) 64 lanes x 4 dwords,
//   row = ns*32*NTW + 32*ntw + (lane&31),  k0 = 128*c + 16*s + 8*(lane>>5) + 2*w       (0 for rows >= N)
//   dst[((((((c*4 + ns)*NTW + ntw)*8 + s)*2 + p)*64 + lane)*4 + w] = {plane_p(W[row][k0 + 1]), plane_p(W[row][k0])}
static __global__ void pack_bslice2h_kernel(const float* __restrict__ src, const float* __restrict__ cs, unsigned* __restrict__ dst, int N, int K, int NTW) {
    const long n = (long)(K / 128) * 4 * NTW * 8 * 2 * 256;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int wd = (int)(e & 3);
        const int lane = (int)((e >> 2) & 63);
        long q = e >> 8;
        const int pl = (int)(q & 1); q >>= 1;
        const int st = (int)(q & 7); q >>= 3;
        const int ntw = (int)(q % NTW); q /= NTW;
        const int ns = (int)(q & 3);
        const int c = (int)(q >> 2);
        const int row = ns * 32 * NTW + 32 * ntw + (lane & 31);
        const int k0 = 128 * c + 16 * st + 8 * (lane >> 5) + 2 * wd;
        unsigned half[2];
        for (int j = 0; j < 2; ++j) {
            const float x = (row < N ? src[(long)row * K + k0 + j] * (cs ? nonzero_gain(cs[k0 + j]) : 1.0f) : 0.0f) * kF16WScale;
            const unsigned h1 = f32_to_f16_bits(x, false);
            half[j] = pl == 0 ? h1 : f32_to_f16_bits(x - f16_bits_to_f32(h1), false);
        }
        dst[e] = half[0] | (half[1] << 16);
    }
}

// Group A of one conv layer: depthwise weight (C,1,k), its bias and the gain / shift (g, b; NULL = 1 / 0) of the LayerNorm that
// feeds the layer ->  taps[j][c] = w[c][j] g[c];  c0[c] = dw_b[c] + sum_j w[c][j] b[c].   (g as packed by copy_gain_kernel)
static __global__ void pack_dw_fold_kernel(const float* __restrict__ w, const float* __restrict__ dwb, const float* __restrict__ g,
                                           const float* __restrict__ b, float* __restrict__ dst, int C, int k) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float gc = g ? nonzero_gain(g[c]) : 1.0f, bc = b ? b[c] : 0.0f;
    float tot = dwb[c];
    for (int j = 0; j < k; ++j) {
        dst[j * C + c] = w[c * k + j] * gc;
        tot += w[c * k + j] * bc;
    }
    dst[k * C + c] = tot;
}
// a LayerNorm gain vector as the kernel uses it (rows outside the sequence are stored as -b/g, see the kernel's store_acc)
static __global__ void copy_gain_kernel(const float* __restrict__ src, float* __restrict__ dst, int n) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) dst[e] = nonzero_gain(src[e]);
}
// h0_pad = LN(tanh(proj_b)) (one row; a single 64-thread block, two-pass in fp32 like the kernels' LayerNorm)
static __global__ void pack_h0_pad_kernel(const float* __restrict__ pb, const float* __restrict__ g, const float* __restrict__ be,
                                          float* __restrict__ dst, int C) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s += tanhf(pb[c]);
    const float mean = s / (float)C;
    float q = 0.0f;
    for (int c = 0; c < C; ++c) { const float d = tanhf(pb[c]) - mean; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(q / (float)C + 1e-5f);
    for (int c = 0; c < C; ++c) dst[c] = fmaf((tanhf(pb[c]) - mean) * rstd, g[c], be[c]);
}
// mel bias folded with the last skip LayerNorm's shift: dst[n] = mel_b[n] + sum_c W[n][c] b_s[c]  (zero for n >= N)
static __global__ void pack_mel_bias_kernel(const float* __restrict__ W, const float* __restrict__ mb, const float* __restrict__ bs,
                                            float* __restrict__ dst, int N, int C, int n_pad) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_pad) return;
    float a = 0.0f;
    if (n < N) {
        a = mb[n];
        for (int c = 0; c < C; ++c) a = fmaf(W[(long)n * C + c], bs[c], a);
    }
    dst[n] = a;
}

struct MelDecP {
    const float* blob;
    DecLayout lay;
    int d4, n_blocks, block_depth, n_mel;
    const float* x;        // (B,T,d4) phoneme-rate (cum != NULL) or (B,L,d4) frame-rate
    const float* h0;       // optional (cum != NULL): (B,T,dx2) = LN(tanh(proj(x))) already computed at PHONEME rate
    const int* cum;        // (B,T) inclusive duration cumsum or NULL
    const int* mel_len;    // (B) or NULL
    const int* lmax_dev;   // device scalar or NULL
    int lmax_host;
    int apply_mask;
    int B, T, L_out;
    float* mel;            // (B, L_out, n_mel)
    int halo, TL;
    int n_tiles;           // windows per utterance
    long long* trace;      // development only (-DESMI_DEC_TRACE): [wave][stamp] shader-clock stamps of block (1,0)
};

template <int DX2>
__host__ __device__ constexpr int dec_lds_floats(int kd) {
    // tile | group A | group B | partial statistics [128][4 column slices][2] (the source-row table of phase 0 aliases them)
    return (kDecRows + 2 * kDecPadRows) * (DX2 + 4) + (kd + 1) * DX2 + 5 * DX2 + kDecRows * 8;
}

// max_b mel_len[b], by every wave for itself: one coalesced read, no extra launch, no atomics; the result is made
// wave-uniform (SGPR) at once.
__device__ __forceinline__ int batch_max_len(const int* __restrict__ mel_len, int B) {
    const int lane = lane_id();
    int v = 0;
    for (int j = lane; j < B; j += 64) v = max(v, mel_len[j]);
    float f = row_max32((float)v);      // lengths are far below 2^24: exact in fp32
    f = fmaxf(f, swap32_f(f));
#ifdef ESMI_WAVESIM
    return (int)f;
#else
    return __builtin_amdgcn_readfirstlane((int)f);
#endif
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// NW waves per window (8): wave (mh = w>>2, ns = w&3) owns rows [128/MH*mh, +128/MH) x columns [ns*DX2/4, +DX2/4).
template <int DX2, int KD, int NW>
__global__ __launch_bounds__(64 * NW, (DX2 <= 128 ? 4 : NW / 4)) void mel_decoder_kernel(const MelDecP p) {
    constexpr int kDecThreads = 64 * NW;    // shadows the namespace constant inside this kernel
    constexpr int NS = 4;                   // column slices per workgroup
    constexpr int MH = NW / NS;             // row groups (2)
    constexpr int MT = 4 / MH;              // 32-row MFMA tiles per wave
    constexpr int NTW = DX2 / (32 * NS);    // 32-column MFMA tiles per wave
    constexpr int WCOLS = 32 * NTW;         // columns per wave
    constexpr int KCH = DX2 / 128;          // 128-channel K chunks of a dx2-wide contraction
    constexpr int LDSROW = DX2 + 4;
    constexpr bool SPLIT = ESMI_DEC_SPLIT == 2;
    constexpr float WSI = SPLIT ? kF16WScaleInv : 1.0f;   // the f16 planes hold 2^8 * W
    constexpr int PAD = KD / 2;
    constexpr int CG = DX2 / 4;             // 4-channel groups per row = threads that share a depthwise strip's rows (32: half a wave, 64: a wave)
    constexpr int RS = kDecRows / (kDecThreads / CG);  // rows per depthwise strip (8 or 16)
    constexpr int WR = RS + 2 * PAD;        // rows of a strip's window
    static_assert(WR <= CG, "one lane of the row group per window row merges its statistics");
    ESMI_DYN_LDS(lds);
    // per-layer parameters in LDS: group A [taps KD*DX2 | c0] (read by the depthwise phase) and group B
    // [pw_b | ln_g | ln_b | skip_g | skip_b] (read on the accumulators).  Single buffers, refilled by LDS-DMA when the last reader of the old
    // contents has passed a barrier: group A of layer l+1 behind layer l's K loop, group B of layer l behind the barrier inside its
    // depthwise phase (the previous block end's skip = LN_s(u) reads the old group B up to that barrier).
    constexpr int GA = (KD + 1) * DX2, GB = 5 * DX2;
    constexpr int P_C0 = KD * DX2;
    constexpr int P_PWB = 0, P_G = DX2, P_B = 2 * DX2, P_SG = 3 * DX2, P_SB = 4 * DX2;   // inside group B
    float* xs = lds;                                                  // [132][LDSROW]
    float* pa = lds + (kDecRows + 2 * kDecPadRows) * LDSROW;          // [GA]
    float* pbuf = pa + GA;                                            // [GB]
    float* pst = pbuf + GB;                                           // [128][NS][2]: (mean, M2) of a row's WCOLS channels held by column slice ns
    int* src = reinterpret_cast<int*>(pst);                           // [128] (phase 0 only)

    const int tid = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const int i = lane & 31, h = lane >> 5;
    const int mh = w / NS, ns = w % NS;
    // XCD-aware workgroup -> (utterance, window) map: workgroup id % 8 is the XCD (round-robin dispatch), so the windows
    // of one utterance are given ids that agree mod 8 and its h0 / cum rows are fetched into ONE XCD's L2 instead of eight.
    int tile, b;
    {
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
    if (f_lo >= valid_end) {  // whole window is padding: the final masked_fill (or the [L, L_out) tail) zeroes it
        const int n = (out_hi - f_lo) * p.n_mel;
        float* o = p.mel + ((long)b * p.L_out + f_lo) * p.n_mel;
        for (int e = tid; e < n; e += kDecThreads) o[e] = 0.0f;
        return;
    }
    const int n_layers = p.n_blocks * p.block_depth;
    // every read of the packed blob is a buffer load: resource + wave-uniform byte offset in SGPRs, one lane-offset VGPR for all of
    // them (64-bit per-lane pointers into the blob, live across the layer loop, were most of the round-2 kernel's register spills)
    const BufRsrc brs = make_rsrc(p.blob, p.lay.total * (long)sizeof(float));
    const unsigned tid16 = (unsigned)tid * 16u, lane16 = (unsigned)lane * 16u;
    auto blob_ld = [&](long float_off, unsigned voff) __attribute__((always_inline)) { return buf_ld4s(brs, voff, (unsigned)(float_off * 4)); };
#ifdef ESMI_DEC_TRACE
    int tr_n = 0;
    const bool tr_on = p.trace && tile == 3 && b == p.B / 2 + 5 && lane == 0;
#define ESMI_STAMP() do { if (tr_on) p.trace[w * 64 + tr_n] = (long long)__builtin_amdgcn_s_memtime(); ++tr_n; } while (0)
#else
#define ESMI_STAMP() do {} while (0)
#endif

    // ---- parameter staging by LDS-DMA (global_load_lds_dwordx4: memory -> LDS, 16 bytes per lane, no staging registers): issued
    // when the last reader of the slots' old contents has passed a barrier, drained by the barrier in front of the first reader of
    // the new ones.
    auto stage = [&](long float_off, float* dst, int n4) __attribute__((always_inline)) {   // n4 float4 from blob + float_off to dst
        for (int base = 0; base < n4; base += kDecThreads) {
            if (base + tid < n4) {
#ifdef ESMI_WAVESIM
                reinterpret_cast<f32x4*>(dst)[base + tid] = blob_ld(float_off + 4 * base, tid16);
#else
                // (opaque: the per-lane source pointer is formed here, not hoisted out of the layer loop as a live 64-bit register pair)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.blob + float_off + 4 * base + opaque_i(4 * tid)),
                                                 (__attribute__((address_space(3))) void*)(dst + 4 * base + 256 * w), 16, 0, 0);
#endif
            }
        }
    };
    auto fetch_A = [&](int l) __attribute__((always_inline)) {   // "layer" n_layers is the mel Linear: its bias goes to group A's first DX2 slots
        if (l < n_layers) stage(l == 0 && p.h0 ? p.lay.layer0_h0 : p.lay.layer0 + (long)l * p.lay.layer_stride, pa, GA / 4);
        else stage(p.lay.mel_b, pa, DX2 / 4);
    };
    auto fetch_B = [&](int l) __attribute__((always_inline)) {
        stage(p.lay.layer0 + (long)l * p.lay.layer_stride + p.lay.l_pwb, pbuf, GB / 4);
    };

    // ---- phase 0: source row of every window row, zero the LDS pad rows, stage proj + layer-0 params
    if (tid < kDecRows) {
        const int f = f0 + tid;
        int s;
        if (f < 0 || f >= L) s = -1;                                   // outside the padded sequence
        else if (p.cum) {
            if (f < mlen) {
                const int ph = frame_to_phoneme(p.cum + b * p.T, p.T, f);
                s = ph < p.T ? b * p.T + ph : -2;
            } else s = -2;                                             // padding frame: zero input row
        } else s = b * L + f;
        src[tid] = s;
    }
    for (int e = tid; e < 2 * kDecPadRows * LDSROW; e += kDecThreads) {
        const int r = e / LDSROW, c = e - r * LDSROW;
        const int rr = r < kDecPadRows ? r : kDecRows + r;             // rows 0,1 and 130,131
        xs[rr * LDSROW + c] = 0.0f;
    }
    if (!p.h0 && tid < 3 * DX2 / 4)                                    // proj_b, proj_g, proj_beta -> group B slots 0..2
        reinterpret_cast<f32x4*>(pbuf)[tid] = blob_ld(p.lay.proj_b, tid16);
    fetch_A(0);
    __syncthreads();

    const bool edge_window = f0 < 0 || f0 + kDecRows > L;   // some window rows lie outside [0, L) (SGPR: a scalar branch)

    f32x16 acc[MT][NTW];
    f32x16 skip[MT][NTW];       // the skip tensor, accumulator layout: lane (i, h) holds frame 32 MT mh + 32 mt + i, channels ns WCOLS + 32 t + 8 g + 4 h + e in [4 g + e]
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[mt][t] = zero16();
        }
    };

    // ================================================================== contractions
    // Un-pipelined form (exact-fp32 build; in-kernel proj stage of the split build): the wave's weight slice for KSUB k-steps
    // is loaded, then the A fragments of its rows stream from LDS.
    constexpr int KSUB = DX2 <= 128 ? (SPLIT ? 4 : 8) / NTW : 8;   // k-steps (of 8 channels) of weights in registers at a time
#if ESMI_DEC_SPLIT
    constexpr int KS16 = KSUB / 2;
    u32x4 bf[NTW][KS16][2];
    auto load_b = [&](long wsl, int k0) __attribute__((always_inline)) {   // wsl: float offset of the wave's weight slice in the blob
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
#pragma unroll
            for (int st = 0; st < KS16; ++st) {
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    bf[t][st][pl] = __builtin_bit_cast(u32x4, blob_ld(wsl + ((t * 8 + (k0 >> 1) + st) * 2 + pl) * 256, lane16));
            }
        }
    };
    // fp32 rows in the tile, split on the fly (esmi_dev.h)
    auto mma_sub = [&](int a_col0, int k0) __attribute__((always_inline)) {
        const float* a_base = xs + opaque_i((kDecPadRows + 32 * MT * mh + i) * LDSROW + 8 * h);
#pragma unroll
        for (int st = 0; st < KS16; ++st) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float* ap = a_base + 32 * mt * LDSROW + a_col0 + 8 * k0 + 16 * st;
                const f16x2p a2 = split_f16x2(*reinterpret_cast<const f32x4*>(ap), *reinterpret_cast<const f32x4*>(ap + 4));
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[mt][t] = mfma32_split2_wx(bf[t][st][0], bf[t][st][1], a2, acc[mt][t]);
            }
        }
    };
    // slice pointer of chunk c of the matrix at float offset `off` (planes: 8 steps x 2 planes x 64 lanes x 16 B per tile)
    auto wslice = [&](long off, int c) __attribute__((always_inline)) { return off + (long)(c * (DX2 / 32) + ns * NTW) * 8 * 2 * 256; };

    // The A operand rows already stored as the two f16 planes (row = [DX2 halves h1 | DX2 halves h2 | pad], written by the
    // depthwise phase): per (16-channel step, row tile) two ds_read_b128 + 3*NTW MFMAs.
    // Two forms.  WD == 0 (dx2 = 128, two workgroups per CU): the weights of KSUB k-steps are loaded, then used -- the compiler
    // keeps one or two fragments in flight and the neighbour workgroup's VALU phases fill the L2 round trips (a ring that
    // pipelines the loop measured slower there: profiles/r03_probes/decoder_round3_experiments.md).  WD > 0 (dx2 = 256, ONE workgroup
    // per CU, nobody to fill the gaps): hand-pipelined -- an item = (16-channel step s, row tile mt); weight fragments of step s + WD
    // and the A fragments of the next item are requested while item q's MFMAs run (VGPR rings; scheduling fences keep hipcc from
    // sinking the loads back to their first use).  small ES decoder 1774 -> 1687 us with WD = 2; WD = 4 spills (1736), 6: 1976.
    constexpr int WD = DX2 > 128 ? ESMI_DEC_WD256 : 0, AD = 1;
    constexpr int NSTEP = 8 * KCH, NITEM = NSTEP * MT;
    static_assert(WD >= 0 && WD <= NSTEP, "ring depth");
    u32x4 wr[WD > 0 ? WD : 1][NTW][2];
    auto w_fetch = [&](long off, int s, int slot) __attribute__((always_inline)) {
        const long wsl = wslice(off, s >> 3);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wr[slot][t][pl] = __builtin_bit_cast(u32x4, blob_ld(wsl + ((t * 8 + (s & 7)) * 2 + pl) * 256, lane16));
        }
    };
    // the first WD weight steps of the matrix at `off` (issued ahead of the barrier that precedes the K loop: the L2 round
    // trip then overlaps the barrier wait)
    auto gemm_prefetch = [&](long off) __attribute__((always_inline)) {
        if constexpr (WD > 0) {
#pragma unroll
            for (int s = 0; s < WD; ++s) w_fetch(off, s, s);
            sched_fence();
        }
    };
    auto gemm_planes = [&](long off) __attribute__((always_inline)) {
        const unsigned* a_base = reinterpret_cast<const unsigned*>(xs) + opaque_i((kDecPadRows + 32 * MT * mh + i) * LDSROW + 4 * h);
        if constexpr (WD == 0) {
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
#pragma unroll
                for (int k0 = 0; k0 < 16; k0 += KSUB) {
                    load_b(wslice(off, c), k0);
#pragma unroll
                    for (int st = 0; st < KS16; ++st) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const unsigned* ap = a_base + 32 * mt * LDSROW + 64 * c + 4 * k0 + 8 * st;
                            f16x2p a2;
                            a2.h1 = *reinterpret_cast<const u32x4*>(ap);
                            a2.h2 = *reinterpret_cast<const u32x4*>(ap + DX2 / 2);
#pragma unroll
                            for (int t = 0; t < NTW; ++t) acc[mt][t] = mfma32_split2_wx(bf[t][st][0], bf[t][st][1], a2, acc[mt][t]);
                        }
                    }
                }
            }
        } else {
            f16x2p ar[AD];
            auto a_fetch = [&](int q, int slot) __attribute__((always_inline)) {
                const int s = q / MT, mt = q % MT;
                const unsigned* ap = a_base + 32 * mt * LDSROW + 64 * (s >> 3) + 8 * (s & 7);
                ar[slot].h1 = *reinterpret_cast<const u32x4*>(ap);
                ar[slot].h2 = *reinterpret_cast<const u32x4*>(ap + DX2 / 2);
            };
#pragma unroll
            for (int q = 0; q < AD; ++q) a_fetch(q, q);
            sched_fence();
#pragma unroll
            for (int q = 0; q < NITEM; ++q) {
                const int s = q / MT, mt = q % MT;
#pragma unroll
                for (int t = 0; t < NTW; ++t)
                    acc[mt][t] = mfma32_split2_wx(wr[s % (WD > 0 ? WD : 1)][t][0], wr[s % (WD > 0 ? WD : 1)][t][1], ar[q % AD], acc[mt][t]);
                if (q + AD < NITEM) a_fetch(q + AD, q % AD);
                if (mt == MT - 1 && s + WD < NSTEP) w_fetch(off, s + WD, s % (WD > 0 ? WD : 1));
                sched_fence();
            }
        }
    };
#else
    f32x4 bf[NTW][KSUB];
    auto load_b = [&](long wsl, int k0) __attribute__((always_inline)) {   // wsl: float offset of the wave's weight slice in the blob
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
#pragma unroll
            for (int kc = 0; kc < KSUB; ++kc) bf[t][kc] = blob_ld(wsl + (t * 16 + k0 + kc) * 256, lane16);
        }
    };
    auto mma_sub = [&](int a_col0, int k0) __attribute__((always_inline)) {
        const float* a_base = xs + opaque_i((kDecPadRows + 32 * MT * mh + i) * LDSROW + 4 * h);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int kc = 0; kc < KSUB; ++kc) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(a_base + 32 * mt * LDSROW + a_col0 + 8 * (k0 + kc));
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int t = 0; t < NTW; ++t) acc[mt][t] = mfma32(bf[t][kc][s], av[s], acc[mt][t]);
                }
            }
        }
    };
    // slice pointer of chunk c of the matrix at float offset `off`
    auto wslice = [&](long off, int c) __attribute__((always_inline)) { return off + (long)(c * (DX2 / 32) + ns * NTW) * 16 * 256; };
    auto gemm_prefetch = [&](long) __attribute__((always_inline)) {};
    auto gemm_planes = [&](long) __attribute__((always_inline)) {};
#endif
    // full dx2-wide contraction over fp32 rows of the tile, un-pipelined
    auto gemm_rows = [&](long off) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += KSUB) {
                load_b(wslice(off, c), k0);
                mma_sub(128 * c, k0);
            }
        }
    };

    // ================================================================== work on the accumulators
    // The products are computed TRANSPOSED (weights as the first MFMA operand): lane (i, h) holds frame i of the row tile and, per
    // register quad g = r >> 2, the four consecutive channels 8g + 4h .. + 3 of every 32-column tile -- one ds_write_b128 per quad.
    // `tanh_acc` turns the accumulators into tanh(acc + bias) IN PLACE; it touches no tile row, so it runs right behind the K loop,
    // before the barrier that waits for the last reader of the operand planes.
    auto tanh_acc = [&](const float* bias) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const float* bp = bias + opaque_i(ns * WCOLS + 32 * t + 4 * h);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bc = *reinterpret_cast<const f32x4*>(bp + 8 * g) * kTanhExpScale;   // the exponent's 2 log2(e) goes into the bias and the scale of the fma
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[mt][t][4 * g + e] = tanh_fast_fma_f32(acc[mt][t][4 * g + e], WSI * kTanhExpScale, bc[e]);
                }
            }
        }
    };
    // partial LayerNorm statistics of the accumulators' rows over this wave's WCOLS channels -> pst[row][ns] = (mean, M2).
    // Two-pass in registers; the two half waves of a row meet through one v_permlane32_swap per pass.
    auto stats_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float s = 0.0f;
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) s += (acc[mt][t][4 * g] + acc[mt][t][4 * g + 1]) + (acc[mt][t][4 * g + 2] + acc[mt][t][4 * g + 3]);
            }
            s += swap32_f(s);
            const float mean = s * (1.0f / WCOLS);
            float q = 0.0f;
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float d = acc[mt][t][e] - mean;
                    q = fmaf(d, d, q);
                }
            }
            q += swap32_f(q);
            if (h == 0) *reinterpret_cast<f32x2*>(pst + (32 * MT * mh + 32 * mt + i) * 8 + 2 * ns) = f32x2{mean, q};
        }
    };
    // merge the four partials of one row (Chan): r = rsqrt(var + eps), m = -mean * r   (nn.LayerNorm: biased variance, eps inside)
    auto merge_row = [&](const float* prow, float& r, float& m) __attribute__((always_inline)) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(prow), c = *reinterpret_cast<const f32x4*>(prow + 4);
        const float mu = ((a[0] + a[2]) + (c[0] + c[2])) * 0.25f;
        const float d0 = a[0] - mu, d1 = a[2] - mu, d2 = c[0] - mu, d3 = c[2] - mu;
        float dd = d0 * d0;
        dd = fmaf(d1, d1, dd);
        dd = fmaf(d2, d2, dd);
        dd = fmaf(d3, d3, dd);
        const float m2 = fmaf((float)WCOLS, dd, (a[1] + a[3]) + (c[1] + c[3]));
        r = rsqrt_fast_f32(fmaf(m2, 1.0f / DX2, 1e-5f));
        m = -mu * r;
    };
    // statistics of this lane's own rows (accumulator layout), merged
    auto merge_own = [&](float (&r)[MT], float (&m)[MT]) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) merge_row(pst + opaque_i((32 * MT * mh + i) * 8) + 32 * mt * 8, r[mt], m[mt]);
    };
    // dst = LN(acc) [+ add] on the accumulators: gain / shift vectors at gp / bp (LDS), row statistics (r, m)
    auto ln_acc = [&](f32x16 (&dst)[MT][NTW], const float* gp, const float* bp, const float (&r)[MT], const float (&m)[MT], auto add_c) __attribute__((always_inline)) {
        constexpr bool ADD = decltype(add_c)::value;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int c0 = opaque_i(ns * WCOLS + 32 * t + 4 * h);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 gg = *reinterpret_cast<const f32x4*>(gp + c0 + 8 * g), bb = *reinterpret_cast<const f32x4*>(bp + c0 + 8 * g);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = fmaf(fmaf(acc[mt][t][4 * g + e], r[mt], m[mt]), gg[e], bb[e]);
                        dst[mt][t][4 * g + e] = ADD ? y + skip[mt][t][4 * g + e] : y;
                    }
                }
            }
        }
    };
    // accumulators -> tile (raw rows).  Rows outside [0, L) (edge windows only: a scalar branch) are stored as -b/g of the LayerNorm
    // (gp, bp) that the rows' consumer folds, with the statistics (1, 0): the consumer's folded taps then see y = g(-b/g) + b = 0 there,
    // which is what the reference's zero padding is.
    auto store_acc = [&](const float* gp, const float* bp) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int c0 = opaque_i(ns * WCOLS + 32 * t + 4 * h);
            float* base = xs + opaque_i((kDecPadRows + 32 * MT * mh + i) * LDSROW + ns * WCOLS + 32 * t + 4 * h);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 q = zero4();
                if (edge_window) {
                    const f32x4 gg = *reinterpret_cast<const f32x4*>(gp + c0 + 8 * g), bb = *reinterpret_cast<const f32x4*>(bp + c0 + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) q[e] = -bb[e] * rcp_fast_f32(gg[e]);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[mt][t][4 * g + e];
                    if (edge_window) {
                        const int f = f0 + 32 * MT * mh + 32 * mt + i;
                        if (f < 0 || f >= L) v = q;
                    }
                    *reinterpret_cast<f32x4*>(base + 32 * mt * LDSROW + 8 * g) = v;
                }
            }
        }
    };
    typedef std::true_type TrueC;
    typedef std::false_type FalseC;

    // ================================================================== the consumer side: normalise on load, depthwise conv, operand planes
    // Thread (cg = tid % CG, strip = tid / CG) owns channels [4cg, +4) of rows [RS strip, +RS).  IDENT: no conv (the mel Linear's
    // operand is the normalised row itself).  UNIT: the rows are already normalised (layer 0 behind a supplied h0): statistics (1, 0).
    auto consume = [&](auto ident_c, bool unit, int l_next_B) __attribute__((always_inline)) {
        constexpr bool IDENT = decltype(ident_c)::value;
        constexpr int HP = IDENT ? 0 : PAD, NR = RS + 2 * HP;
        const int tid_o = opaque_i(tid);         // (the strip indices are re-derived here: hoisted out of the layer loop they were spilled)
        const int dw_cg = tid_o % CG, dw_r0 = (tid_o / CG) * RS;
        {   // lane j of the row group merges window row j's statistics -> the row's pad floats (neighbouring groups write the
            // rows they share with identical bits).  Rows outside [0, L) hold -b/g and the statistics (1, 0) (store_acc); rows
            // outside the window are the zero pad rows: (0, 0).
            const int j = lane % CG;
            if (j < NR) {
                const int pr = dw_r0 - HP + j, f = f0 + pr;
                float r = 0.0f, m = 0.0f;
                if (pr >= 0 && pr < kDecRows) {
                    if (unit || f < 0 || f >= L) r = 1.0f;
                    else merge_row(pst + pr * 8, r, m);
                }
                *reinterpret_cast<f32x2*>(xs + (kDecPadRows + pr) * LDSROW + DX2) = f32x2{r, m};
            }
        }
        // The operand planes are written IN PLACE.  Inside a strip that is safe without a barrier: its CG lanes are one (half) wave,
        // a wave's LDS operations execute in order, and row p's raw values are read (by all lanes) in an instruction that precedes the
        // one that writes row p's planes.  Between strips only the halo rows are shared: they go to registers ahead of the barrier;
        // the strip's own rows are streamed behind it, a few rows ahead of their use.
        f32x4 win[NR];
        const float* col = xs + opaque_i((kDecPadRows + dw_r0 - HP) * LDSROW + 4 * dw_cg);
        const float* stp = xs + opaque_i((kDecPadRows + dw_r0 - HP) * LDSROW + DX2);
        unsigned* prow = reinterpret_cast<unsigned*>(xs) + opaque_i((kDecPadRows + dw_r0) * LDSROW + 2 * dw_cg);
        auto ld_row = [&](int q) __attribute__((always_inline)) { win[q] = *reinterpret_cast<const f32x4*>(col + q * LDSROW); };
        auto norm_row = [&](int q) __attribute__((always_inline)) {
            const f32x2 st = *reinterpret_cast<const f32x2*>(stp + q * LDSROW);
#pragma unroll
            for (int e = 0; e < 4; ++e) win[q][e] = fmaf(win[q][e], st[0], st[1]);
        };
        constexpr int AHEAD = 3;                     // own rows in flight ahead of the row being produced
        if constexpr (!IDENT) {
#pragma unroll
            for (int q = 0; q < HP; ++q) { ld_row(q); ld_row(NR - 1 - q); }
            ESMI_STAMP();   // 1: statistics merged, halo rows requested
            __syncthreads();
            ESMI_STAMP();   // 2: barrier
        } else {
            lds_wave_sync();
        }
        if (l_next_B >= 0) fetch_B(l_next_B);      // the old group B (a block end's skip LN) was last read ahead of this barrier
        f32x4 tap[IDENT ? 1 : KD];       // in registers for all RS rows
        f32x4 tb = zero4();
        const float* pat = pa + opaque_i(4 * dw_cg);
        if constexpr (!IDENT) {
#pragma unroll
            for (int j = 0; j < KD; ++j) tap[j] = *reinterpret_cast<const f32x4*>(pat + j * DX2);
            tb = *reinterpret_cast<const f32x4*>(pat + P_C0);
        }
        // own rows q = HP .. HP + RS - 1; output row r needs window rows r .. r + 2 HP
#pragma unroll
        for (int q = HP; q < HP + RS && q < 2 * HP + 1 + AHEAD; ++q) ld_row(q);
#pragma unroll
        for (int q = 0; q < HP; ++q) { norm_row(q); norm_row(NR - 1 - q); }
#pragma unroll
        for (int q = HP; q < HP + RS && q < 2 * HP + 1; ++q) norm_row(q);
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            f32x4 a;
            if constexpr (IDENT) a = win[r];
            else {
                a = tb;
#pragma unroll
                for (int j = 0; j < KD; ++j) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = fmaf(win[r + j][e], tap[j][e], a[e]);
                }
            }
            wave_lockstep();   // every lane of the strip holds the raw rows up to r + 2 HP + AHEAD: row r may be overwritten
            if (SPLIT) {   // the K loop's A operand, already split (esmi_dev.h): 4 channels = 2 dwords per plane
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                unsigned h1a, h2a, h1b, h2b;
                split_f16_pair(a[0], a[1], h1a, h2a);
                split_f16_pair(a[2], a[3], h1b, h2b);
                unsigned* rowp = prow + r * LDSROW;
                *reinterpret_cast<u32x2*>(rowp) = u32x2{h1a, h1b};
                *reinterpret_cast<u32x2*>(rowp + DX2 / 2) = u32x2{h2a, h2b};
            } else {
                *reinterpret_cast<f32x4*>(const_cast<float*>(col) + (r + HP) * LDSROW) = a;
            }
            // next rows: raw row (r + 2 HP + 1 + AHEAD) is requested now, row (r + 2 HP + 1) is normalised for the next output
            if (r + 2 * HP + 1 + AHEAD < HP + RS) ld_row(r + 2 * HP + 1 + AHEAD);
            if (r + 2 * HP + 1 < HP + RS) norm_row(r + 2 * HP + 1);
        }
    };

    // ---- first stage: Linear(d4, dx2) + Tanh + LN.  All three are row-wise, and a frame's input row is its phoneme's row: when
    // the caller supplies h0 = LN(tanh(proj(x))) at PHONEME rate (enc_fuse_va_kernel computes it while the features
    // are still on the CU) the stage reduces to a gather -- one of the six GEMM stages of the window disappears
    // (D frames per phoneme share one row).  Padding frames (zero input rows) get LN(tanh(proj_b)), packed as `h0_pad`.
    if (p.h0) {
        for (int e = tid; e < kDecRows * (DX2 / 4); e += kDecThreads) {
            const int r = e / (DX2 / 4), q = e - r * (DX2 / 4);
            const int s = src[r];
            f32x4 v = zero4();
            if (s >= 0) v = ld4(p.h0 + (long)s * DX2 + 4 * q);
            else if (s == -2) v = blob_ld(p.lay.h0_pad, (unsigned)(16 * q));
            *reinterpret_cast<f32x4*>(xs + (kDecPadRows + r) * LDSROW + 4 * q) = v;
        }
        __syncthreads();
        // skip = the stage's output, accumulator layout
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const float* base = xs + opaque_i((kDecPadRows + 32 * MT * mh + i) * LDSROW + ns * WCOLS + 32 * t + 4 * h);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(base + 32 * mt * LDSROW + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) skip[mt][t][4 * g + e] = v[e];
                }
            }
        }
    } else {
        zero_acc();
        const int nchunks = p.d4 / 128;
        for (int ch = 0; ch < nchunks; ++ch) {
            if (ch > 0) __syncthreads();  // previous chunk fully consumed
            for (int e = tid; e < kDecRows * 32; e += kDecThreads) {
                const int r = e >> 5, q = e & 31;
                const int s = src[r];
                f32x4 v = zero4();
                if (s >= 0) v = ld4(p.x + (long)s * p.d4 + ch * 128 + 4 * q);
                *reinterpret_cast<f32x4*>(xs + (kDecPadRows + r) * LDSROW + 4 * q) = v;
            }
            __syncthreads();
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += KSUB) {
                load_b(wslice(p.lay.proj_w, ch), k0);
                mma_sub(0, k0);
            }
        }
        tanh_acc(pbuf + P_PWB);
        __syncthreads();  // every wave finished reading the staged input (and the source-row table the statistics overwrite)
        stats_acc();
        store_acc(pbuf + P_G, pbuf + P_B);
        __syncthreads();
        {   // skip = LN(tanh(proj)); the tile keeps the raw rows, layer 0's taps carry the gain / shift
            float r[MT], m[MT];
            merge_own(r, m);
            ln_acc(skip, pbuf + P_G, pbuf + P_B, r, m, FalseC{});
        }
    }

    // ---- conv layers
    for (int l = 0; l < n_layers; ++l) {
        const long lbase = p.lay.layer0 + (long)l * p.lay.layer_stride;
        const bool block_end = ((l + 1) % p.block_depth) == 0;
        ESMI_STAMP();   // 0: layer start
        // A. normalise on load + depthwise conv in place -> the K loop's operand planes
        consume(FalseC{}, l == 0 && p.h0 != nullptr, l);
        gemm_prefetch(lbase + p.lay.l_pw);   // first weight steps: in flight across the barrier
        ESMI_STAMP();   // 3: dw written
        __syncthreads();
        ESMI_STAMP();   // 4: barrier
        // K. pointwise conv: K = dx2;  bias + tanh and the partial statistics on the accumulators (no tile access: ahead of the barrier)
        zero_acc();
        if (SPLIT) gemm_planes(lbase + p.lay.l_pw);
        else gemm_rows(lbase + p.lay.l_pw);
        ESMI_STAMP();   // 5: K loop issued
        tanh_acc(pbuf + P_PWB);
        stats_acc();
        fetch_A(l + 1);      // next layer's taps (their slots were last read by this layer's depthwise phase)
        ESMI_STAMP();   // 6: tanh + statistics
        __syncthreads();     // all reads of the operand planes done; partials visible
        ESMI_STAMP();   // 7: barrier
        if (block_end) {     // end of a decoder block: u = LN(t) + skip (networks.py:299); the tile gets u, the next consumer LN_s's statistics
            float r[MT], m[MT];
            merge_own(r, m);
            __syncthreads();  // (the partials' slots are reused for u's)
            ln_acc(acc, pbuf + P_G, pbuf + P_B, r, m, TrueC{});
            stats_acc();
            store_acc(pbuf + P_SG, pbuf + P_SB);
        } else {
            store_acc(pbuf + P_G, pbuf + P_B);
        }
        ESMI_STAMP();   // 8: rows stored (block end: + LN, skip add, statistics, one more barrier)
        __syncthreads();
        ESMI_STAMP();   // 9: barrier
        if (block_end && l + 1 < n_layers) {   // skip = LN_s(u), accumulator layout (the last block's only consumer is the mel Linear: folded)
            float r[MT], m[MT];
            merge_own(r, m);
            ln_acc(skip, pbuf + P_SG, pbuf + P_SB, r, m, FalseC{});
        }
    }

    // ---- mel Linear(dx2, n_mel) on skip = LN_s(u): the tile holds u and its statistics, the matrix carries LN_s's gain / shift
    consume(TrueC{}, false, -1);
    gemm_prefetch(p.lay.mel_w);
    __syncthreads();
    if (ns * WCOLS < p.n_mel) {   // wave-uniform: column slices beyond n_mel have nothing to do
        zero_acc();
        if (SPLIT) gemm_planes(p.lay.mel_w);
        else gemm_rows(p.lay.mel_w);
        const float* mb = pa;                        // folded mel bias (zero padded to dx2)
        const bool vec_ok = (p.n_mel & 3) == 0;      // rows of 16-byte multiples: float4 stores
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int f = f0 + 32 * MT * mh + 32 * mt + i;
                if (f < f_lo || f >= out_hi) continue;
                float* orow = p.mel + ((long)b * p.L_out + f) * p.n_mel;
                const bool live = f < valid_end;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = ns * WCOLS + 32 * t + 8 * g + 4 * h;
                    if (col >= p.n_mel) continue;
                    const f32x4 bc = *reinterpret_cast<const f32x4*>(mb + col);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = live ? fmaf(acc[mt][t][4 * g + e], WSI, bc[e]) : 0.0f;
                    if (vec_ok) {
                        *reinterpret_cast<f32x4*>(orow + col) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (col + e < p.n_mel) orow[col + e] = v[e];
                    }
                }
            }
        }
    }
}

}  // namespace esmi
