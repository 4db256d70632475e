// Fully fused mel decoder -- MelDecoder.forward, layers/networks.py:291-304, plus the length
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
// per side = the depthwise conv's in-window padding; +4 floats/row keep the 16-byte row-fragment
// reads bank-conflict free).
//
// Weight-stationary GEMMs (ESMI_DEC_SPLIT: exact-fp32 MFMA, or fp32-accurate split products on the bf16 / f16 matrix pipe).  Wave w = (mh = w>>2, ns = w&3)
// owns rows [64mh, 64mh+64) x columns [ns*DX2/4, +DX2/4).  For each 128-channel K chunk it loads
// its weight slice ONCE into registers (16 coalesced 16-byte loads per 32-column tile, from the
// pre-packed blob) and streams the A fragments of its 64 rows from LDS: the K loop touches no
// global memory.  (Round-1 ablation: with one wave owning 32 full rows, re-streaming the weights
// from L2 for every 32 rows cost 150 us of a 675 us kernel.)
//
// Per conv layer, five short phases separated by workgroup barriers:
//   1. depthwise k-tap conv IN PLACE on the tile (each thread: 4 channels x 8 or 16 rows, window in
//      registers);  2. K loop (MFMA only + ds_read_b128);  3. bias + tanh, accumulators -> tile;
//   4. LayerNorm by row-owner threads (16 threads = one DPP row per tile row, 4 rows per thread so that the gain / shift
//      vectors are read once per 4 rows; the rows' values and the skip tensor in registers; block end: LN_s(x + skip));
//      rows outside [0, L) are forced to 0.
//
// Fidelity notes (SURVEY.md §7 "hard parts"):
//  * frames in [mel_len[b], L) are PADDING FRAMES: zero input rows, but computed like any other frame,
//    because the reference computes them and the k-tap conv leaks them into the last valid frames;
//  * frames outside [0, L) do not exist in the reference: every layer's Conv1d zero-pads there, so
//    such rows are forced to 0 after every LayerNorm;
//  * rows >= mel_len[b] of the output are zeroed only at the very end (the final masked_fill).
#pragma once
#include "esmi_dev.h"
#include "small_kernels.h"

#ifdef ESMI_ABL_NO_TANH
#define ESMI_DEC_TANH(x) (x)
#endif
#ifndef ESMI_DEC_TANH
#define ESMI_DEC_TANH tanh_fast_f32
#endif
// Build knobs of the dx2 = 128 instantiation (measured on MI355X, tiny ES B=256 T=128, decoder time inside bench.py; the three
// lines below are from the exact-fp32 build with the proj stage still at frame rate.  Since then, same workload:
// proj at phoneme rate 0.41 ms, split-bf16x3 contraction 0.31 ms, split-f16x2 0.24 ms, 16-lane LayerNorm rows 0.215 ms;
// marginal costs by ablation at the 0.24 ms point: LayerNorm 56 us, operand split 29 us, MFMA 15 us, tanh 9 us):
//   WPS=2 KSUB=16 LOWREG=0 : 235 VGPRs, no spill, ONE workgroup per CU ........ 0.500 ms
//   WPS=3 KSUB=16 LOWREG=1 : 168 VGPRs, no spill, one workgroup per CU ........ 0.505 ms
//   WPS=4 KSUB=8  LOWREG=1 : 128 VGPRs, 55 spilled, TWO workgroups per CU ..... 0.477 ms   <- default
//     (the spill is mostly the 32-register skip tensor, dead during the K loops; it shows as ~390 MB of scratch
//      traffic per launch in the PMC counters -- profiles/r01_e_pmc_counters.json)
// Two co-resident workgroups hide part of each other's non-MFMA phases; the gain is small because the resident
// partner's K loop starves the other's LayerNorm phase (phase traces: 3k -> 11-19k cycles).
#ifndef ESMI_DEC_RSQRT
#define ESMI_DEC_RSQRT rsqrt_fast_f32   // v_rsq_f32 (1 ulp); every LayerNorm thread computes it for 4 rows
#endif
#ifndef ESMI_DEC_WPS16
#define ESMI_DEC_WPS16 8      // waves/SIMD of the 16-wave-per-window build (ESMI_DEC_NW128=16): 8 = two workgroups per CU at 64 VGPRs
#endif
#ifndef ESMI_DEC_LN_TPR
#define ESMI_DEC_LN_TPR 16    // LayerNorm threads per row: 16 (4 rows per thread, gain/shift read once per 4 rows), 8 or 4
#endif
#ifndef ESMI_DEC_FUSED_LN
#define ESMI_DEC_FUSED_LN 0   // (measured: 0.2085 vs 0.1928 ms, i.e. slower, kept as an option) bias + tanh + LayerNorm (+ block-end skip LayerNorm) on the accumulators in registers: every wave
                              // reduces its column slice of a row in-lane, the four slices' (mean, M2) meet in a small LDS
                              // table (Chan merge); the activations are stored once instead of store / load / store
#endif
#ifndef ESMI_DEC_TAPS_IN_REGS
#define ESMI_DEC_TAPS_IN_REGS 1
#endif
#ifndef ESMI_DEC_PRESPLIT
#define ESMI_DEC_PRESPLIT 1   // split-f16x2 only: the depthwise phase writes its output rows as the two f16 planes (same bytes as
                              // fp32, in place), so the K loop's A fragments need no conversion (4 waves read every element)
#endif
#ifndef ESMI_DEC_WPS
#define ESMI_DEC_WPS 4      // __launch_bounds__ waves/SIMD (512-thread workgroups: 2 -> 256 VGPRs, 4 -> 128 VGPRs)
#endif
#ifndef ESMI_DEC_LOWREG
#define ESMI_DEC_LOWREG 1   // 1: no cross-phase prefetch (weights, taps, params fetched where used): fewer live registers
#endif
#ifndef ESMI_DEC_CHAIN_PRIO
#define ESMI_DEC_CHAIN_PRIO 0   // wave priority during the non-MFMA phases (exact-fp32 build: +1 % with two workgroups per CU; split-f16 build: -1 %)
#endif
#if defined(ESMI_WAVESIM)
#define ESMI_PRIO(n) do {} while (0)
#else
#define ESMI_PRIO(n) do { if (ESMI_DEC_CHAIN_PRIO) __builtin_amdgcn_s_setprio(n); } while (0)
#endif
#ifndef ESMI_DEC_NS
#define ESMI_DEC_NS 4      // column slices per workgroup of the dx2 = 128 build: 4 (two row groups) or 2 (four row groups: half the
                           // LDS A-fragment reads but twice the weight sub-slices per K loop: measured 0.522 vs 0.484 ms)
#endif
#ifndef ESMI_DEC_SPLIT      // contraction of the pointwise GEMMs (esmi_dev.h):
#define ESMI_DEC_SPLIT 2    //   0: v_mfma_f32_32x32x2_f32 (exact fp32)      1: fp32 split into 3 bf16, 6 products on v_mfma_f32_32x32x16_bf16
#endif                      //   2: fp32 split into 2 f16 (weights pre-scaled by 2^8), 3 products on v_mfma_f32_32x32x16_f16
#ifndef ESMI_DEC_KSUB
#define ESMI_DEC_KSUB (ESMI_DEC_SPLIT ? 4 : 8)   // k-steps (of 8 channels) of the weight slice held in registers at a time (16 = all of
                                                 // K = 128); split-bf16 path: 4 -> 24 VGPRs of planes (measured: 2: 0.319, 4: 0.309-0.317, 8: 0.323 ms)
#endif

namespace esmi {

constexpr int kDecRows = 128;     // frames per workgroup window
constexpr int kDecPadRows = 2;    // zero rows above/below the window in LDS (>= k/2)
constexpr int kDecThreads = 512;  // 8-wave windows (NW = 8, the default for every dx2)
constexpr int kMelCols = 96;      // n_mel <= 96 (three 32-column MFMA tiles)

struct DecLayout {  // offsets in floats into the packed blob
    long proj_w, proj_b, proj_g, proj_beta;
    long layer0, layer_stride;                 // per conv layer
    long l_dw, l_dwb, l_pw, l_pwb, l_g, l_b;   // relative to the layer base
    long skip0;                                // per block: gain[dx2], bias[dx2]
    long mel_w, mel_b;
    long total;
};

inline DecLayout dec_layout(int d4, int dx2, int kd, int n_blocks, int block_depth) {
    DecLayout L;
    long o = 0;
    constexpr long kWNum = ESMI_DEC_SPLIT == 1 ? 3 : 2;   // matrix storage: three bf16 planes (1.5x), two f16 planes or fp32
    L.proj_w = o; o += (long)d4 * dx2 * kWNum / 2;
    L.proj_b = o; o += dx2;                    // proj_b, proj_g, proj_beta contiguous
    L.proj_g = o; o += dx2;
    L.proj_beta = o; o += dx2;
    L.l_dw = 0;                                // per layer: taps[kd][dx2], dw_b, pw_b, ln_g, ln_b contiguous ...
    L.l_dwb = (long)kd * dx2;
    L.l_pwb = L.l_dwb + dx2;
    L.l_g = L.l_pwb + dx2;
    L.l_b = L.l_g + dx2;
    L.l_pw = L.l_b + dx2;                      // ... then the packed pointwise matrix
    L.layer_stride = L.l_pw + (long)dx2 * dx2 * kWNum / 2;
    L.layer0 = o; o += L.layer_stride * n_blocks * block_depth;
    L.skip0 = o; o += 2L * dx2 * n_blocks;
    L.mel_w = o; o += (long)dx2 * dx2 * kWNum / 2;   // packed like a dx2 x dx2 matrix, rows >= n_mel zero
    L.mel_b = o; o += dx2;                     // zero padded
    L.total = o;
    return L;
}

// Weight-stationary B-fragment packing of a (N, K) row-major matrix, K a multiple of 128, for a
// workgroup whose 4 column slices are WCOLS = 32*NTW wide:
//   dst[(((((c*4 + ns)*NTW + ntw)*16 + kc)*64 + lane)*4 + s] =
//       W[ns*WCOLS + 32*ntw + (lane&31)][128*c + 8*kc + 4*(lane>>5) + s]      (0 for rows >= N)
__global__ void pack_bslice_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int K, int NTW) {
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
        dst[e] = row < N ? src[(long)row * K + col] : 0.0f;
    }
}

// The same slices as three bf16 planes (hi / mid / lo by truncation, esmi_dev.h) in the B layout of
// v_mfma_f32_32x32x16_bf16: per (chunk c, column slice ns, tile ntw, 16-channel step s, plane p) 64 lanes x 4 dwords:
//   dst[((((((c*4 + ns)*NTW + ntw)*8 + s)*3 + p)*64 + lane)*4 + w] = {plane_p(W[row][k0 + 1]), plane_p(W[row][k0])},
//   row = ns*32*NTW + 32*ntw + (lane&31),  k0 = 128*c + 16*s + 8*(lane>>5) + 2*w       (0 for rows >= N)
__global__ void pack_bslice3_kernel(const float* __restrict__ src, unsigned* __restrict__ dst, int N, int K, int NTW) {
    const long n = (long)(K / 128) * 4 * NTW * 8 * 3 * 256;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int wd = (int)(e & 3);
        const int lane = (int)((e >> 2) & 63);
        long q = e >> 8;
        const int pl = (int)(q % 3); q /= 3;
        const int st = (int)(q & 7); q >>= 3;
        const int ntw = (int)(q % NTW); q /= NTW;
        const int ns = (int)(q & 3);
        const int c = (int)(q >> 2);
        const int row = ns * 32 * NTW + 32 * ntw + (lane & 31);
        const int k0 = 128 * c + 16 * st + 8 * (lane >> 5) + 2 * wd;
        unsigned half[2];
        for (int j = 0; j < 2; ++j) {
            const float x = row < N ? src[(long)row * K + k0 + j] : 0.0f;
            const unsigned h = __builtin_bit_cast(unsigned, x) & 0xFFFF0000u;
            const float r1 = x - __builtin_bit_cast(float, h);
            const unsigned m = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
            const float r2 = r1 - __builtin_bit_cast(float, m);
            half[j] = pl == 0 ? h : (pl == 1 ? m : (__builtin_bit_cast(unsigned, r2) & 0xFFFF0000u));
        }
        dst[e] = half[1] | (half[0] >> 16);
    }
}

// ... and as two binary16 planes of 2^8 * W (round to nearest; esmi_dev.h) in the same layout with 2 planes per step:
//   dst[((((((c*4 + ns)*NTW + ntw)*8 + s)*2 + p)*64 + lane)*4 + w] = {plane_p(W[row][k0 + 1]), plane_p(W[row][k0])}
__global__ void pack_bslice2h_kernel(const float* __restrict__ src, unsigned* __restrict__ dst, int N, int K, int NTW) {
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
            const float x = (row < N ? src[(long)row * K + k0 + j] : 0.0f) * kF16WScale;
            const unsigned h1 = f32_to_f16_bits(x, false);
            half[j] = pl == 0 ? h1 : f32_to_f16_bits(x - f16_bits_to_f32(h1), false);
        }
        dst[e] = half[0] | (half[1] << 16);
    }
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
    return (kDecRows + 2 * kDecPadRows) * (DX2 + 4) + (kd + 6) * DX2 + kDecRows + (ESMI_DEC_FUSED_LN ? kDecRows * 8 : 0);
}

// NW = waves per window.  NW = 8: wave (mh = w>>2, ns = w&3) owns 64 rows x DX2/4 columns, one workgroup per CU.
// NW = 4: wave ns owns all 128 rows x DX2/4 columns; the workgroup fits twice on a CU (76 KB LDS), which would let
// one workgroup's tanh / LayerNorm / depthwise phases run under the other's K loop -- but with ROCm 7.2's hipcc the
// 4-wave build needs 256 VGPRs + 212 spilled and is slower (700 vs 560 us); kept selectable for the next round.
// (Also tried and dropped in round 1: two windows per workgroup in explicit ping-pong -- correct, 330-450 spills.)
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

template <int DX2, int KD, int NW>
__global__ __launch_bounds__(64 * NW, (DX2 <= 128 && NW == 8 ? ESMI_DEC_WPS : (NW == 16 ? ESMI_DEC_WPS16 : 2))) void mel_decoder_kernel(const MelDecP p) {
    constexpr int kDecThreads = 64 * NW;    // shadows the namespace constant inside this kernel
    constexpr int NS = (DX2 <= 128 && NW == 8) ? ESMI_DEC_NS : 4;   // column slices per workgroup
    constexpr int MH = NW / NS;             // row groups (1, 2 or 4)
    constexpr int MT = 4 / MH;              // 32-row MFMA tiles per wave (4, 2 or 1)
    constexpr int TPR = ESMI_DEC_LN_TPR;          // LayerNorm threads per row (adjacent lanes inside one DPP row: the statistics are DPP adds)
    constexpr int RPT = kDecRows * TPR / kDecThreads;   // rows per LayerNorm thread (4 or 8): the gain/shift vectors of its
                                                  // channels are read from LDS once for all of them
    constexpr bool LOWREG = ESMI_DEC_LOWREG && DX2 <= 128;
    constexpr int NTW = DX2 / (32 * NS);    // 32-column MFMA tiles per wave
    constexpr int WCOLS = 32 * NTW;         // columns per wave
    constexpr int KCH = DX2 / 128;          // 128-channel K chunks of a dx2-wide contraction
    constexpr int LDSROW = DX2 + 4;
    constexpr float WSI = ESMI_DEC_SPLIT == 2 ? kF16WScaleInv : 1.0f;   // the f16 planes hold 2^8 * W
    constexpr int PAD = KD / 2;
    constexpr int CG = DX2 / 4;             // 4-channel groups per row
    constexpr int RS = kDecRows / (kDecThreads / CG);  // rows per depthwise strip (8 or 16)
    constexpr int NV = DX2 / (4 * TPR);     // float4 per LayerNorm thread and row (channels 4*c + 4*TPR*k, c = lane % TPR)
    ESMI_DYN_LDS(lds);
    // per-layer small parameters in LDS: [taps KD*DX2 | dw_b] (group A: read by the depthwise phase) and
    // [pw_b | ln_g | ln_b | skip_g | skip_b] (group B: read by the tanh / LayerNorm phases).  Single buffer:
    // layer l+1's group A is committed during layer l's tanh phase, its group B during layer l+1's
    // depthwise phase -- each when no reader of the old contents is left.
    constexpr int PB = (KD + 6) * DX2;
    constexpr int P_DWB = KD * DX2, P_PWB = P_DWB + DX2, P_G = P_PWB + DX2, P_B = P_G + DX2, P_SG = P_B + DX2,
                  P_SB = P_SG + DX2;
    constexpr int NA4 = (KD + 1) * DX2 / 4;                            // float4 in group A
    constexpr int NB4 = 3 * DX2 / 4;                                   // float4 of pw_b, ln_g, ln_b
    static_assert(NA4 <= 2 * kDecThreads && NB4 + DX2 / 2 <= kDecThreads, "param staging: one float4 per thread per group");
    float* xs = lds;                                                  // [132][LDSROW]
    float* pbuf = lds + (kDecRows + 2 * kDecPadRows) * LDSROW;        // [PB]
    int* src = reinterpret_cast<int*>(pbuf + PB);                     // [128]
    float* stats = pbuf + PB + kDecRows;                              // [128][4 column slices][mean, M2]   (ESMI_DEC_FUSED_LN)

    const int tid = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const int i = lane & 31, h = lane >> 5;
    const int mh = w / NS, ns = w % NS;     // NW = 4: mh == 0
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
    const f32x4* blob4 = reinterpret_cast<const f32x4*>(p.blob);
#ifdef ESMI_DEC_TRACE
    int tr_n = 0;
    const bool tr_on = p.trace && tile == 3 && b == p.B / 2 + 5 && lane == 0;
#define ESMI_STAMP() do { if (tr_on) p.trace[w * 64 + tr_n] = (long long)__builtin_amdgcn_s_memtime(); ++tr_n; } while (0)
#else
#define ESMI_STAMP() do {} while (0)
#endif

    // ---- parameter staging: global -> register (issued early) ... register -> LDS (committed later).
    // "layer" n_layers is the mel Linear (group B = its bias only).
    f32x4 pstA = zero4(), pstA2 = zero4(), pstB = zero4();
    auto issue_A = [&](int l) __attribute__((always_inline)) {
        if (LOWREG) return;
        if (l < n_layers && tid < NA4) pstA = blob4[((p.lay.layer0 + (long)l * p.lay.layer_stride) >> 2) + tid];
        if (l < n_layers && tid + kDecThreads < NA4)
            pstA2 = blob4[((p.lay.layer0 + (long)l * p.lay.layer_stride) >> 2) + tid + kDecThreads];
    };
    auto commit_A = [&](int l) __attribute__((always_inline)) {
        if (LOWREG) {
            if (l < n_layers)
                for (int e = tid; e < NA4; e += kDecThreads)
                    reinterpret_cast<f32x4*>(pbuf)[e] = blob4[((p.lay.layer0 + (long)l * p.lay.layer_stride) >> 2) + e];
            return;
        }
        if (l < n_layers && tid < NA4) reinterpret_cast<f32x4*>(pbuf)[tid] = pstA;
        if (l < n_layers && tid + kDecThreads < NA4) reinterpret_cast<f32x4*>(pbuf)[tid + kDecThreads] = pstA2;
    };
    auto issue_B_now = [&](int l) __attribute__((always_inline)) {
        if (l < n_layers) {
            if (tid < NB4) pstB = blob4[((p.lay.layer0 + (long)l * p.lay.layer_stride + p.lay.l_pwb) >> 2) + tid];
            else if (((l + 1) % p.block_depth) == 0 && tid < NB4 + DX2 / 2)   // block end: skip LN params
                pstB = blob4[((p.lay.skip0 + (long)(l / p.block_depth) * 2 * DX2) >> 2) + tid - NB4];
        } else if (tid < DX2 / 4) {
            pstB = blob4[(p.lay.mel_b >> 2) + tid];
        }
    };
    auto issue_B = [&](int l) __attribute__((always_inline)) {
        if (!LOWREG) issue_B_now(l);
    };
    auto commit_B = [&](int l) __attribute__((always_inline)) {
        if (LOWREG) issue_B_now(l);
        f32x4* d4 = reinterpret_cast<f32x4*>(pbuf + P_PWB);
        if (l < n_layers) {
            if (tid < NB4 || (((l + 1) % p.block_depth) == 0 && tid < NB4 + DX2 / 2)) d4[tid] = pstB;
        } else if (tid < DX2 / 4) {
            d4[tid] = pstB;
        }
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
    if (tid < NB4)                                                     // proj_b, proj_g, proj_beta -> group B
        reinterpret_cast<f32x4*>(pbuf + P_PWB)[tid] = blob4[(p.lay.proj_b >> 2) + tid];
    issue_A(0);
    commit_A(0);
    __syncthreads();

    // LayerNorm ownership: lane = 16*rg + c; the wave owns rows [4*RPT*w, +4*RPT), the thread rows ln_row0 + (0..RPT-1)
    // and the float4 channel groups c + 16*k of each
    const int ln_c = lane & (TPR - 1), ln_row0 = (64 / TPR) * RPT * w + RPT * (lane / TPR);
    unsigned ln_inside = 0;                 // bit j: row ln_row0 + j exists in the reference (inside [0, L))
#pragma unroll
    for (int j = 0; j < RPT; ++j) ln_inside |= (src[ln_row0 + j] != -1 ? 1u : 0u) << j;

    f32x16 acc[MT][NTW];
    f32x4 skip[RPT][NV];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
#pragma unroll
        for (int v = 0; v < NV; ++v) skip[j][v] = zero4();
    }

    // weight-stationary GEMM pieces.  bf = this wave's weight slice for KSUB k-steps (64 VGPRs); it is
    // (re)loaded right after the previous K loop so the L2 latency hides under the non-MFMA phases.
    constexpr int KSUB = DX2 <= 128 ? ESMI_DEC_KSUB / NTW : 8;   // k-steps of weights in registers at a time (32 / 64 VGPRs)
#if ESMI_DEC_SPLIT
    // split contraction (esmi_dev.h): per 16-channel step one A fragment (8 fp32 from the tile, split on the fly) against the
    // NPL pre-split weight planes; KSUB/2 steps of weights (NPL x 4 VGPRs each per tile) in registers at a time
    constexpr int KS16 = KSUB / 2;
    constexpr int NPL = ESMI_DEC_SPLIT == 1 ? 3 : 2;
    u32x4 bf[NTW][KS16][NPL];
    auto load_b = [&](const f32x4* wsl, int k0) __attribute__((always_inline)) {
        const u32x4* w3 = reinterpret_cast<const u32x4*>(wsl);
#ifdef ESMI_ABL_NOWLOAD
        if (p.n_mel >= 0) return;
#endif
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
#pragma unroll
            for (int st = 0; st < KS16; ++st) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) bf[t][st][pl] = w3[((t * 8 + (k0 >> 1) + st) * NPL + pl) * 64];
            }
        }
    };
    auto mma_sub = [&](int a_col0, int k0) __attribute__((always_inline)) {
        const float* a_base = xs + opaque_i((kDecPadRows + 32 * MT * mh + i) * LDSROW + 8 * h);
#pragma unroll
        for (int st = 0; st < KS16; ++st) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float* ap = a_base + 32 * mt * LDSROW + a_col0 + 8 * k0 + 16 * st;
#if ESMI_DEC_SPLIT == 1
                const bf16x3 a3 = split_bf16x3(*reinterpret_cast<const f32x4*>(ap), *reinterpret_cast<const f32x4*>(ap + 4));
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[mt][t] = mfma32_split_wx(bf[t][st][0], bf[t][st][1], bf[t][st][2], a3, acc[mt][t]);
#else
#if defined(ESMI_ABL_NOSPLIT)
                f16x2p a2;
                a2.h1 = __builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(ap));
                a2.h2 = __builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(ap + 4));
#elif defined(ESMI_ABL_NOAREAD)
                f16x2p a2;
                a2.h1 = u32x4{(unsigned)st, (unsigned)mt, (unsigned)lane, 3u};
                a2.h2 = a2.h1;
#else
                const f16x2p a2 = split_f16x2(*reinterpret_cast<const f32x4*>(ap), *reinterpret_cast<const f32x4*>(ap + 4));
#endif
#ifdef ESMI_ABL_NOMFMA
                if (p.n_mel >= 0) { acc[mt][0][0] += __builtin_bit_cast(float, a2.h1[0] ^ a2.h1[1] ^ a2.h1[2] ^ a2.h1[3] ^ a2.h2[0] ^ a2.h2[1] ^ a2.h2[2] ^ a2.h2[3]); continue; }
#endif
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[mt][t] = mfma32_split2_wx(bf[t][st][0], bf[t][st][1], a2, acc[mt][t]);
#endif
            }
        }
    };
    // slice pointer of chunk c of the matrix at float offset `off` (planes: 8 steps x NPL planes x 64 lanes x 16 B per tile)
    auto wslice = [&](long off, int c) __attribute__((always_inline)) { return blob4 + (off >> 2) + (long)(c * (DX2 / 32) + ns * NTW) * 8 * NPL * 64 + lane; };
#if ESMI_DEC_SPLIT == 2 && ESMI_DEC_PRESPLIT
    // the same with the A rows already stored as planes by the depthwise phase: row = [DX2 halves h1 | DX2 halves h2 | pad]
    auto mma_sub_pre = [&](int a_col0, int k0) __attribute__((always_inline)) {
        const unsigned* a_base = reinterpret_cast<const unsigned*>(xs) + opaque_i((kDecPadRows + 32 * MT * mh + i) * LDSROW + 4 * h);
#pragma unroll
        for (int st = 0; st < KS16; ++st) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned* ap = a_base + 32 * mt * LDSROW + (a_col0 >> 1) + 4 * k0 + 8 * st;
                f16x2p a2;
                a2.h1 = *reinterpret_cast<const u32x4*>(ap);
                a2.h2 = *reinterpret_cast<const u32x4*>(ap + DX2 / 2);
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[mt][t] = mfma32_split2_wx(bf[t][st][0], bf[t][st][1], a2, acc[mt][t]);
            }
        }
    };
    constexpr bool PRESPLIT = true;
#else
    auto mma_sub_pre = [&](int, int) __attribute__((always_inline)) {};
    constexpr bool PRESPLIT = false;
#endif
#else
    f32x4 bf[NTW][KSUB];
    auto load_b = [&](const f32x4* wsl, int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
#pragma unroll
            for (int kc = 0; kc < KSUB; ++kc) bf[t][kc] = wsl[(t * 16 + k0 + kc) * 64];
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
    auto mma_sub_pre = [&](int, int) __attribute__((always_inline)) {};
    constexpr bool PRESPLIT = false;
    // slice pointer of chunk c of the matrix at float offset `off`
    auto wslice = [&](long off, int c) __attribute__((always_inline)) { return blob4 + (off >> 2) + (long)(c * (DX2 / 32) + ns * NTW) * 16 * 64 + lane; };
#endif
    // full dx2-wide contraction with the first sub-slice already in bf; leaves `next`'s first sub-slice in bf
    auto gemm_dx2 = [&](long off, const f32x4* next, bool pre) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += KSUB) {
                if (LOWREG || c > 0 || k0 > 0) load_b(wslice(off, c), k0);
                if (pre) mma_sub_pre(128 * c, k0);
                else mma_sub(128 * c, k0);
            }
        }
        if (next && !LOWREG) load_b(next, 0);
    };
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[mt][t] = zero16();
        }
    };
    // accumulators (+ bias, tanh) -> tile.  The products are computed TRANSPOSED (weights as the first MFMA operand): lane
    // (i, h) holds frame i of the tile and, per register quad g = r >> 2, the four consecutive channels 8g + 4h .. + 3 --
    // one ds_write_b128 per quad instead of four ds_write_b32 (and float4 global stores for the mel rows).
    auto store_tanh = [&](const float* bias) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int col = ns * WCOLS + 32 * t + 4 * h;
            const float* bp = bias + opaque_i(col);
            float* base = xs + opaque_i((kDecPadRows + 32 * MT * mh + i) * LDSROW + col);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 bc = *reinterpret_cast<const f32x4*>(bp + 8 * g);
#if !defined(ESMI_ABL_NO_TANH)
                bc *= kTanhExpScale;     // the exponent's 2 log2(e) goes into the bias and the scale of the fma
#endif
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#if defined(ESMI_ABL_NO_TANH)
                        v[e] = fmaf(acc[mt][t][4 * g + e], WSI, bc[e]);
#else
                        v[e] = tanh_fast_fma_f32(acc[mt][t][4 * g + e], WSI * kTanhExpScale, bc[e]);
#endif
                    }
                    *reinterpret_cast<f32x4*>(base + 32 * mt * LDSROW + 8 * g) = v;
                }
            }
        }
    };
    // LayerNorm of one row: this thread's NV float4 of it in v[], gain / shift of the same channels in g[] / be[]
    // (two-pass; 16 threads per row)
    auto ln_regs = [&](f32x4 (&v)[NV], const f32x4 (&g)[NV], const f32x4 (&be)[NV]) __attribute__((always_inline)) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k) s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
        s = row_sum_n<TPR>(s);
        const float mean = s * (1.0f / DX2);
        float q = 0.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[k][e] - mean;
                q = fmaf(d, d, q);
            }
        }
        q = row_sum_n<TPR>(q);
        const float rstd = ESMI_DEC_RSQRT(q * (1.0f / DX2) + 1e-5f);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[k][e] = fmaf((v[k][e] - mean) * rstd, g[k][e], be[k][e]);
        }
    };
    auto ln_params = [&](const float* pv, f32x4 (&o)[NV]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NV; ++k) o[k] = *reinterpret_cast<const f32x4*>(pv + 4 * TPR * k);
    };
    // LN pass over the tile (in place): x = LN(x) [; x = LN_s(x + skip)] ; outside rows -> 0 ; skip update
    auto ln_pass = [&](const float* pb0, bool block_end, bool set_skip) __attribute__((always_inline)) {
        const float* pb = pb0 + opaque_i(4 * ln_c);           // this thread's channels of every param vector
        float* ln_ptr = xs + opaque_i((kDecPadRows + ln_row0) * LDSROW + 4 * ln_c);
        f32x4 g[NV], be[NV];
        ln_params(pb + P_G, g);
        ln_params(pb + P_B, be);
        f32x4 v[RPT][NV];
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
#pragma unroll
            for (int k = 0; k < NV; ++k) v[j][k] = *reinterpret_cast<const f32x4*>(ln_ptr + j * LDSROW + 4 * TPR * k);
        }
#ifndef ESMI_ABL_NO_LN
#pragma unroll
        for (int j = 0; j < RPT; ++j) ln_regs(v[j], g, be);
        if (block_end) {  // end of a decoder block: skip = LN_s(x + skip), networks.py:299
            ln_params(pb + P_SG, g);
            ln_params(pb + P_SB, be);
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
#pragma unroll
                for (int k = 0; k < NV; ++k) v[j][k] += skip[j][k];
                ln_regs(v[j], g, be);
            }
        }
#endif
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (!((ln_inside >> j) & 1u)) v[j][k] = zero4();
                *reinterpret_cast<f32x4*>(ln_ptr + j * LDSROW + 4 * TPR * k) = v[j][k];
                if (set_skip) skip[j][k] = v[j][k];
            }
        }
    };

    // ---- fused epilogue (ESMI_DEC_FUSED_LN).  In the transposed accumulator layout lane (i, h) of wave (mh, ns) holds, for
    // each of its MT row tiles, row 32*MT*mh + 32*mt + i and the channels ns*WCOLS + 32t + 8g + 4h + (0..3): half of the
    // wave's column slice of that row, the other half sits in lane i + 32.  A row's LayerNorm statistics are therefore
    // in-lane sums + one v_permlane32_swap per wave, and a 2-float LDS entry per (row, slice) to meet the other slices.
    f32x16 skp[MT][NTW];         // the skip tensor in that layout
    unsigned acc_inside = 0;     // bit mt: the lane's row of tile mt exists in the reference (inside [0, L))
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc_inside |= (src[32 * MT * mh + 32 * mt + i] != -1 ? 1u : 0u) << mt;
    const int ep_col = ns * WCOLS + 4 * h;      // + 32t + 8g
    float* ep_stats = stats + opaque_i((32 * MT * mh + i) * 2 * NS);   // + 64*NS*mt: this lane's row entry [NS][2]
    float* ep_tile = xs + opaque_i((kDecPadRows + 32 * MT * mh + i) * LDSROW + ep_col);
    // x = LN(x) over the full rows; gp / bp = the gain / shift vectors in LDS.  Contains one workgroup barrier.
    auto ln_exchange = [&](f32x16 (&x)[MT][NTW], const float* gp, const float* bp) __attribute__((always_inline)) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {   // two-pass statistics of this wave's WCOLS channels of the row
            float sm = 0.0f;
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) sm += (x[mt][t][4 * g] + x[mt][t][4 * g + 1]) + (x[mt][t][4 * g + 2] + x[mt][t][4 * g + 3]);
            }
            sm += swap32_f(sm);
            const float mean = sm * (1.0f / WCOLS);
            float q = 0.0f;
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = x[mt][t][r] - mean;
                    q = fmaf(d, d, q);
                }
            }
            q += swap32_f(q);
            *reinterpret_cast<f32x2*>(ep_stats + 64 * NS * mt + 2 * ns) = f32x2{mean, q};   // both half-wave lanes: same value
        }
        __syncthreads();
        float mean[MT], rstd[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {   // Chan merge of the NS equal-sized slices (identical in every wave)
            f32x2 st[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) st[k] = *reinterpret_cast<const f32x2*>(ep_stats + 64 * NS * mt + 2 * k);
            float m = 0.0f;
#pragma unroll
            for (int k = 0; k < NS; ++k) m += st[k][0];
            m *= 1.0f / NS;
            float m2 = 0.0f;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const float d = st[k][0] - m;
                m2 += fmaf((float)WCOLS * d, d, st[k][1]);
            }
            mean[mt] = m;
            rstd[mt] = ESMI_DEC_RSQRT(m2 * (1.0f / DX2) + 1e-5f);
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 gg = *reinterpret_cast<const f32x4*>(gp + ep_col + 32 * t + 8 * g);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bp + ep_col + 32 * t + 8 * g);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[mt][t][4 * g + e] = fmaf((x[mt][t][4 * g + e] - mean[mt]) * rstd[mt], gg[e], bb[e]);
                }
            }
        }
    };
    // bias + tanh + LN [+ LN_s(x + skip)] on the accumulators, rows outside the sequence -> 0, result -> tile (and skip)
    auto epilogue_ln = [&](const float* pb, bool block_end, int l_next) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bc = *reinterpret_cast<const f32x4*>(pb + P_PWB + ep_col + 32 * t + 8 * g);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[mt][t][4 * g + e] = ESMI_DEC_TANH(fmaf(acc[mt][t][4 * g + e], WSI, bc[e]));
                }
            }
        }
        commit_A(l_next);                    // next layer's taps (their LDS slots were last read by this layer's depthwise phase)
        issue_B(l_next);
        ln_exchange(acc, pb + P_G, pb + P_B);       // its barrier also ends the K loop's reads of the tile
        if (block_end) {   // end of a decoder block: skip = LN_s(x + skip), networks.py:299
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[mt][t] += skp[mt][t];
            }
            __syncthreads();                 // every wave has read the first table
            ln_exchange(acc, pb + P_SG, pb + P_SB);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                if (!((acc_inside >> mt) & 1u)) acc[mt][t] = zero16();
                if (block_end) skp[mt][t] = acc[mt][t];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[mt][t][4 * g + e];
                    *reinterpret_cast<f32x4*>(ep_tile + 32 * mt * LDSROW + 32 * t + 8 * g) = v;
                }
            }
        }
    };

    // ---- proj: Linear(d4, dx2) + Tanh + LN.  All three are row-wise, and a frame's input row is its phoneme's row: when
    // the caller supplies h0 = LN(tanh(proj(x))) at PHONEME rate (enc_fuse_va_kernel computes it while the features
    // are still on the CU) the stage reduces to a gather -- one of the six GEMM stages of the window disappears
    // (D frames per phoneme share one row).  Padding frames (zero input rows) get LN(tanh(proj_b)).
    if (p.h0) {
        for (int e = tid; e < kDecRows * (DX2 / 4); e += kDecThreads) {
            const int r = e / (DX2 / 4), q = e - r * (DX2 / 4);
            const int s = src[r];
            f32x4 v = zero4();
            if (s >= 0) v = ld4(p.h0 + (long)s * DX2 + 4 * q);
            else if (s == -2) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(pbuf + P_PWB + 4 * q);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = ESMI_DEC_TANH(bb[c]);
            }
            *reinterpret_cast<f32x4*>(xs + (kDecPadRows + r) * LDSROW + 4 * q) = v;
        }
        if (!LOWREG) load_b(n_layers > 0 ? wslice(p.lay.layer0 + p.lay.l_pw, 0) : wslice(p.lay.mel_w, 0), 0);
        issue_B(0);
        __syncthreads();
        {   // row owners: LayerNorm only for the padding frames' rows; skip = the stage's output
            const float* pb = pbuf + opaque_i(4 * ln_c);
            float* ln_ptr = xs + opaque_i((kDecPadRows + ln_row0) * LDSROW + 4 * ln_c);
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
#pragma unroll
                for (int k = 0; k < NV; ++k) skip[j][k] = *reinterpret_cast<const f32x4*>(ln_ptr + j * LDSROW + 4 * TPR * k);
            }
            unsigned pad_rows = 0;
#pragma unroll
            for (int j = 0; j < RPT; ++j) pad_rows |= (src[ln_row0 + j] == -2 ? 1u : 0u) << j;
            if (ballot64(pad_rows != 0u) != 0ull) {   // wave-uniform: the row reductions inside are wave-level exchanges
                f32x4 g[NV], be[NV];
                ln_params(pb + P_G, g);
                ln_params(pb + P_B, be);
#pragma unroll
                for (int j = 0; j < RPT; ++j) {
                    f32x4 u[NV];
#pragma unroll
                    for (int k = 0; k < NV; ++k) u[k] = skip[j][k];
                    ln_regs(u, g, be);
#pragma unroll
                    for (int k = 0; k < NV; ++k) {
                        if ((pad_rows >> j) & 1u) skip[j][k] = u[k];
                        *reinterpret_cast<f32x4*>(ln_ptr + j * LDSROW + 4 * TPR * k) = skip[j][k];
                    }
                }
            }
        }
        __syncthreads();
    } else {
    zero_acc();
    const int nchunks = p.d4 / 128;
    if (!LOWREG) load_b(wslice(p.lay.proj_w, 0), 0);
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
            if (LOWREG || ch > 0 || k0 > 0) load_b(wslice(p.lay.proj_w, ch), k0);
            mma_sub(0, k0);
        }
    }
    // weights of the first conv layer (or of the mel Linear) start flowing while proj's epilogue runs
    if (!LOWREG) load_b(n_layers > 0 ? wslice(p.lay.layer0 + p.lay.l_pw, 0) : wslice(p.lay.mel_w, 0), 0);
    __syncthreads();  // every wave finished reading the staged input
    issue_B(0);
    store_tanh(pbuf + P_PWB);
    __syncthreads();
    ln_pass(pbuf, false, true);
    __syncthreads();

    }

    if (ESMI_DEC_FUSED_LN) {   // the stage's output (in the tile) is the first skip tensor
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(ep_tile + 32 * mt * LDSROW + 32 * t + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) skp[mt][t][4 * g + e] = v[e];
                }
            }
        }
    }

    // ---- conv layers
    const int dw_cg = tid % CG, dw_r0 = (tid / CG) * RS;
    for (int l = 0; l < n_layers; ++l) {
        const float* pb = pbuf;
        const long lbase = p.lay.layer0 + (long)l * p.lay.layer_stride;
        ESMI_STAMP();   // 0: layer start
        // 1. depthwise conv in place: window -> registers | barrier | filtered rows -> tile
        {
            f32x4 win[RS + 2 * PAD];
            float* col = xs + opaque_i((kDecPadRows + dw_r0 - PAD) * LDSROW + 4 * dw_cg);
            const float* pbt = pb + opaque_i(4 * dw_cg);
            unsigned* prow = reinterpret_cast<unsigned*>(xs) + opaque_i((kDecPadRows + dw_r0) * LDSROW + 2 * dw_cg);
#pragma unroll
            for (int r = 0; r < RS + 2 * PAD; ++r) win[r] = *reinterpret_cast<const f32x4*>(col + r * LDSROW);
            f32x4 tap[KD];       // in registers for all RS rows: re-reading them per row cost 8 x KD ds_read_b128 per thread
            if (ESMI_DEC_TAPS_IN_REGS) {
#pragma unroll
                for (int j = 0; j < KD; ++j) tap[j] = *reinterpret_cast<const f32x4*>(pbt + j * DX2);
            }
            const f32x4 tb = *reinterpret_cast<const f32x4*>(pbt + P_DWB);
            ESMI_STAMP();   // 1: window loaded (issued)
            __syncthreads();
            ESMI_STAMP();   // 2: barrier passed
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                f32x4 a = tb;
#pragma unroll
                for (int j = 0; j < KD; ++j) {
#ifdef ESMI_ABL_NO_DW
                    if (j != PAD) continue;
#endif
                    const f32x4 tj = ESMI_DEC_TAPS_IN_REGS ? tap[j] : *reinterpret_cast<const f32x4*>(pbt + j * DX2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = fmaf(win[r + j][e], tj[e], a[e]);
                }
                if (PRESPLIT) {   // the K loop's A operand, already split (esmi_dev.h): 4 channels = 2 dwords per plane
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    unsigned h1a, h2a, h1b, h2b;
                    split_f16_pair(a[0], a[1], h1a, h2a);
                    split_f16_pair(a[2], a[3], h1b, h2b);
                    unsigned* rowp = prow + r * LDSROW;
                    *reinterpret_cast<u32x2*>(rowp) = u32x2{h1a, h1b};
                    *reinterpret_cast<u32x2*>(rowp + DX2 / 2) = u32x2{h2a, h2b};
                } else {
                    *reinterpret_cast<f32x4*>(col + (r + PAD) * LDSROW) = a;
                }
            }
        }
        commit_B(l);                         // this layer's bias / LN params (issued during the previous layer)
        issue_A(l + 1);                      // next layer's taps: in flight during the K loop
        ESMI_STAMP();   // 3: dw written
        __syncthreads();
        ESMI_STAMP();   // 4: barrier
        // 2. pointwise conv: K = dx2, weights register-stationary; then prefetch the next matrix's first slice
        ESMI_PRIO(0);
        zero_acc();
        gemm_dx2(lbase + p.lay.l_pw, l + 1 < n_layers ? wslice(lbase + p.lay.layer_stride + p.lay.l_pw, 0)
                                                      : wslice(p.lay.mel_w, 0), PRESPLIT);
        ESMI_STAMP();   // 5: K loop issued
        ESMI_PRIO(ESMI_DEC_CHAIN_PRIO);
        const bool block_end = ((l + 1) % p.block_depth) == 0;
        if (ESMI_DEC_FUSED_LN) {
            ESMI_STAMP();   // 6
            ESMI_STAMP();   // 7
            ESMI_STAMP();   // 8
            epilogue_ln(pb, block_end, l + 1);
            ESMI_STAMP();   // 9: epilogue done
            __syncthreads();
            ESMI_STAMP();   // 10: barrier
        } else {
            __syncthreads();  // all reads of the filtered tile done
            ESMI_STAMP();   // 6: barrier
            // 3. bias + tanh -> tile; commit the staged params of layer l+1 to the other buffer
            store_tanh(pb + P_PWB);
            commit_A(l + 1);
            issue_B(l + 1);
            ESMI_STAMP();   // 7: tanh stored
            __syncthreads();
            ESMI_STAMP();   // 8: barrier
            // 4. LayerNorm (+ block-end skip LayerNorm) by row owners
            ln_pass(pb, block_end, block_end);
            ESMI_STAMP();   // 9: LN done
            __syncthreads();
            ESMI_STAMP();   // 10: barrier
        }
    }

    // ---- mel Linear(dx2, n_mel) on skip (held in the tile), masked store
    if (n_layers == 0) issue_B(0);
    commit_B(n_layers);    // mel bias; the last LayerNorm's reads of group B finished before its closing barrier
    __syncthreads();
    if (ns * WCOLS < p.n_mel) {   // wave-uniform: column slices beyond n_mel have nothing to do
        zero_acc();
        gemm_dx2(p.lay.mel_w, nullptr, false);
        const float* mb = pbuf + P_PWB;
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
