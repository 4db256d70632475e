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
#include <type_traits>

#include "esmi_dev.h"
#include "small_kernels.h"

// Build knob: the contraction form (two libraries of one ABI are built from it, __graft_entry__.py)
#ifndef ESMI_DEC_SPLIT      // contraction of the pointwise GEMMs (esmi_dev.h):
#define ESMI_DEC_SPLIT 2    //   0: v_mfma_f32_32x32x2_f32 (exact fp32; the libesmi_fp32mfma.so build)
#endif                      //   2: fp32 split into 2 f16 (weights pre-scaled by 2^8), 3 products on v_mfma_f32_32x32x16_f16
#if ESMI_DEC_SPLIT != 0 && ESMI_DEC_SPLIT != 2
#error "ESMI_DEC_SPLIT must be 0 (fp32 MFMA) or 2 (split f16x2)"
#endif
// Fixed choices (each the measured best on MI355X; the alternatives and their times are in HISTORY.md 3.1, not in the build):
//  * dx2 = 128: 4 waves per SIMD (128 VGPRs, two workgroups per CU); weight slices of 4 k-steps (split build) / 8 (fp32 build) loaded then
//    used, no hand-pipelined ring (a ring measured 169 -> 172 / 174 us: the neighbour workgroup already fills the L2 round trips);
//  * dx2 = 256 (one workgroup per CU, nobody fills the gaps): the K loop's weight fragments run 2 steps ahead in a VGPR ring
//    (small ES decoder 1774 -> 1687 us; depth 4 spills: 1736), the A fragments one item ahead; 8 waves per window (16: 2.43 vs 1.87 ms).
constexpr int kDecWps128 = 4, kDecWeightRing256 = 2, kDecARing = 1;
#ifndef ESMI_DEC_MEL_NT     // dx2 = 256: the chunk's mel rows leave as streaming (non-temporal) stores
#define ESMI_DEC_MEL_NT 1
#endif
#ifndef ESMI_DEC_H0_NT      // dx2 = 256: the h0 row gather as streaming loads
#define ESMI_DEC_H0_NT 1    // (base ES, B = 512: HBM traffic per launch 1051 MB with plain stores and loads, 576 MB with streaming mel stores,
#endif                      //  541 MB with both, against 386 MB algorithmic: profiles/r06_probes/decoder256_traffic_ab.txt)
#ifndef ESMI_DEC_CUM_LDS    // the utterance's duration scan is copied into LDS (one round trip) and the frame -> phoneme search runs there
#define ESMI_DEC_CUM_LDS 1  // (0: a binary search in global memory, log2(T) dependent L2 round trips per chunk: profiles/r06_dec_budget.md)
#endif
#ifndef ESMI_DEC_LN_ACC     // dx2 = 256: LayerNorm on the K loop's accumulators -- per-(row, column slice) sums through a 1 KB-per-32-rows LDS exchange,
#define ESMI_DEC_LN_ACC 1   // normalised in registers, stored once (0: tanh rows -> tile -> barrier -> row owners read, normalise, write back)
#endif
#define ESMI_DEC_TANH tanh_fast_f32
#define ESMI_DEC_RSQRT rsqrt_fast_f32   // v_rsq_f32 (1 ulp)

namespace esmi {

constexpr int kDecRows = 128;     // frames per workgroup window
constexpr int kDecPadRows = 2;    // zero rows above/below the window in LDS (>= k/2)
constexpr int kDecThreads = 512;  // 8-wave windows (the dx2 = 128 kernel and the host-side launch default)
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
    constexpr long kWNum = 2;   // matrix storage in units of DX2*DX2/2 floats: two f16 planes or fp32 (the same bytes)
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
static __global__ void pack_bslice_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int K, int NTW) {
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

// The same slices as two binary16 planes of 2^8 * W (round to nearest; esmi_dev.h) in the B layout of v_mfma_f32_32x32x16_f16:
// per (chunk c, column slice ns, tile ntw, 16-channel step s, plane p) 64 lanes x 4 dwords,
//   row = ns*32*NTW + 32*ntw + (lane&31),  k0 = 128*c + 16*s + 8*(lane>>5) + 2*w       (0 for rows >= N)
//   dst[((((((c*4 + ns)*NTW + ntw)*8 + s)*2 + p)*64 + lane)*4 + w] = {plane_p(W[row][k0 + 1]), plane_p(W[row][k0])}
static __global__ void pack_bslice2h_kernel(const float* __restrict__ src, unsigned* __restrict__ dst, int N, int K, int NTW) {
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

// The same layout as the kernel sees it: everything inside a conv layer is a compile-time offset (DX2, KD are template parameters), and
// the few run-time offsets are 32-bit (the blob is a few MB).  The kernel used to take the 16 64-bit fields of DecLayout as arguments:
// 32 SGPRs live across the layer loop, which is where the dx2 = 256 kernel's SGPR spills (48) and its uniform values in VGPRs came from.
template <int DX2, int KD>
struct DecLay {
    static constexpr int l_dwb = KD * DX2, l_pwb = l_dwb + DX2, l_g = l_pwb + DX2, l_b = l_g + DX2, l_pw = l_b + DX2,
                         layer_stride = l_pw + DX2 * DX2;
    int proj_b, layer0, skip0, mel_w, mel_b, total;
    __host__ __device__ DecLay(int d4, int n_blocks, int block_depth) {
        proj_b = d4 * DX2;
        layer0 = proj_b + 3 * DX2;
        skip0 = layer0 + layer_stride * n_blocks * block_depth;
        mel_w = skip0 + 2 * DX2 * n_blocks;
        mel_b = mel_w + DX2 * DX2;
        total = mel_b + DX2;
    }
    // (the host's dec_layout() is the one the packer writes by; the launcher checks that the two agree)
    bool matches(const DecLayout& L) const {
        return L.proj_w == 0 && L.proj_b == proj_b && L.proj_g == proj_b + DX2 && L.proj_beta == proj_b + 2 * DX2 && L.layer0 == layer0 &&
               L.layer_stride == layer_stride && L.l_dw == 0 && L.l_dwb == l_dwb && L.l_pwb == l_pwb && L.l_g == l_g && L.l_b == l_b &&
               L.l_pw == l_pw && L.skip0 == skip0 && L.mel_w == mel_w && L.mel_b == mel_b && L.total == total;
    }
};

struct MelDecP {
    const float* blob;
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
    int halo;              // rows a window loses per side without carried state: (k/2) * conv layers
    int seg_len;           // frames per segment (a workgroup's share of an utterance)
    int n_seg;             // segments per utterance
    float* carry_ws;       // dx2 = 256 with multi-chunk segments: per workgroup `ws_stride` floats of scratch ([conv layer slot][k/2][dx2],
                           // then [block boundary][kDecBlockCarry4 float4]), else NULL
    int ws_stride;
    int carry_lds_layers;  // conv-layer carry slots kept in LDS behind the tile (what fits); the rest live in `carry_ws`
    int skew;              // dx2 = 256 chunk walk: the tile's frame base steps back by block_depth * k/2 rows at every block boundary
                           // (a chunk then loses block_depth * k/2 rows on its right instead of the whole halo), see the chunk loop
    long long* trace;      // development only (-DESMI_DEC_TRACE): [wave][stamp] shader-clock stamps of block (1,0)
};
// measurement aid (esmi_mel_decoder_clock_probe, include/esmi.h): two {shader clock, 100 MHz clock} stamps per launch, see the chunk
// loop.  A device global per translation unit (like the range flag), not a kernel argument: the kernel is at its register limit.
ESMI_DEVICE_GLOBAL_PTR(long long, g_dec_clk);
static inline int store_dec_clock_pointer(long long* slots) { return ESMI_STORE_DEVICE_GLOBAL_PTR(g_dec_clk, slots); }

// block skew: a block boundary hands (block_depth + 1) * k/2 rows of dx2 floats to the next chunk, one float4 per thread
constexpr int kDecBlockCarry4 = 512;
// conv layers whose carried rows (k/2 rows of dx2 floats each) fit in LDS behind the tile and the parameter slots (dx2 = 256 only)
template <int DX2>
__host__ __device__ constexpr int dec_lds_floats(int kd);
template <int DX2>
inline int dec_carry_lds_layers(int kd, int n_layers) {
    if (DX2 <= 128) return 0;
    const int free_f = 160 * 1024 / 4 - dec_lds_floats<DX2>(kd), per = (kd / 2) * DX2;
    const int n = free_f / per;
    return n < n_layers ? n : n_layers;
}
template <int DX2>
__host__ __device__ constexpr int dec_lds_floats(int kd) {
    return (kDecRows + 2 * kDecPadRows) * (DX2 + 4) + (kd + 6) * DX2 + kDecRows + (ESMI_DEC_LN_ACC && DX2 > 128 ? 2 * kDecRows * 8 : 0);
}

// max_b mel_len[b], by every wave for itself: one coalesced read, no extra launch, no atomics; the result is made
// wave-uniform (SGPR) at once.
__device__ __forceinline__ int batch_max_len(const int* __restrict__ mel_len, int B) {
    const int lane = lane_id();
    int v = 0;
    for (int j = lane; j < B; j += 64) v = max(v, mel_len[j]);
    float f = row_max32((float)v);      // lengths are far below 2^24: exact in fp32
    f = fmaxf(f, swap32_f(f));
    return uniform_i((int)f);
}

// NW waves per window (8 or 16): wave (mh = w>>2, ns = w&3) owns rows [128/MH*mh, +128/MH) x columns [ns*DX2/4, +DX2/4).
// NW = 16 (dx2 = 256, one workgroup per CU either way): four waves per SIMD instead of two inside every barrier-separated
// phase -- the phases are latency-bound, so the extra waves are what hides it -- at 128 VGPRs (one 32-row tile per wave).
template <int DX2, int KD, int NW>
__global__ __launch_bounds__(64 * NW, (DX2 <= 128 ? kDecWps128 : NW / 4)) void mel_decoder_kernel(const MelDecP p) {
    constexpr int kDecThreads = 64 * NW;    // shadows the namespace constant inside this kernel
    constexpr int NS = 4;                   // column slices per workgroup
    constexpr int MH = NW / NS;             // row groups (2 or 4)
    constexpr int MT = 4 / MH;              // 32-row MFMA tiles per wave
    constexpr int TPR = 16;                 // LayerNorm threads per row (one DPP row: the statistics are 4 DPP adds)
    constexpr int RPT = kDecRows * TPR / kDecThreads;   // rows per LayerNorm thread (4): gain / shift vectors are read once for all of them
    constexpr int NTW = DX2 / (32 * NS);    // 32-column MFMA tiles per wave
    constexpr int WCOLS = 32 * NTW;         // columns per wave
    constexpr int KCH = DX2 / 128;          // 128-channel K chunks of a dx2-wide contraction
    constexpr int LDSROW = DX2 + 4;
    constexpr bool SPLIT = ESMI_DEC_SPLIT == 2;
    constexpr float WSI = SPLIT ? kF16WScaleInv : 1.0f;   // the f16 planes hold 2^8 * W
    constexpr int PAD = KD / 2;
    constexpr int CG = DX2 / 4;             // 4-channel groups per row
    constexpr int RS = kDecRows / (kDecThreads / CG);  // rows per depthwise strip (8 or 16)
    constexpr int NV = DX2 / (4 * TPR);     // float4 per LayerNorm thread and row (channel groups c + 16k, c = lane % 16)
    ESMI_DYN_LDS(lds);
    // per-layer small parameters in LDS: [taps KD*DX2 | dw_b] (group A: read by the depthwise phase) and
    // [pw_b | ln_g | ln_b | skip_g | skip_b] (group B: read by the tanh / LayerNorm phases).  Single buffer: layer l+1's
    // group A is fetched at the start of layer l's tanh phase and committed at its end, group B of layer l during layer l's
    // depthwise phase -- each when no reader of the old contents is left; the global-memory latency hides behind the phase.
    constexpr int PB = (KD + 6) * DX2;
    constexpr int P_DWB = KD * DX2, P_PWB = P_DWB + DX2, P_G = P_PWB + DX2, P_B = P_G + DX2, P_SG = P_B + DX2,
                  P_SB = P_SG + DX2;
    constexpr int NA4 = (KD + 1) * DX2 / 4;                            // float4 in group A
    constexpr int NB4 = 3 * DX2 / 4;                                   // float4 of pw_b, ln_g, ln_b
    static_assert(NA4 <= kDecThreads && NB4 + DX2 / 2 <= kDecThreads, "param staging: one float4 per thread per group");
    float* xs = lds;                                                  // [132][LDSROW]
    float* pbuf = lds + (kDecRows + 2 * kDecPadRows) * LDSROW;        // [PB]
    int* src = reinterpret_cast<int*>(pbuf + PB);                     // [128]

    int tid = (int)threadIdx.x, lane = lane_id();       // (re-derived per chunk below: see the chunk loop)
    const int w = uniform_i(wave_id());  // an SGPR: everything derived from the wave's place (mh, ns, weight-slice offsets) stays scalar
    int i = lane & 31, h = lane >> 5;
    const int mh = w / NS, ns = w % NS;
    // XCD-aware workgroup -> (utterance, window) map: workgroup id % 8 is the XCD (round-robin dispatch), so the windows
    // of one utterance are given ids that agree mod 8 and its h0 / cum rows are fetched into ONE XCD's L2 instead of eight.
    int seg, b;
    {
        const int id = (int)blockIdx.x, per8 = 8 * p.n_seg;
        const int g = id / per8, r = id - g * per8;
        seg = r >> 3;
        b = 8 * g + (r & 7);
        if (b >= p.B) return;
    }
    const int L = p.lmax_dev ? *p.lmax_dev : (p.lmax_host >= 0 ? p.lmax_host : batch_max_len(p.mel_len, p.B));
    const int mlen = p.mel_len ? min(p.mel_len[b], L) : L;
    const int valid_end = p.apply_mask ? mlen : L;
    // The workgroup's segment [s0, s1) of the utterance, walked in chunks of one 128-row tile.  The first chunk of a segment that
    // does not start the utterance recomputes `halo` rows on its left (nothing to carry in from); every chunk loses `halo` rows on
    // its right.  STREAM (dx2 = 256): consecutive chunks advance by 128 - halo frames and each conv layer's k/2 input rows in front
    // of the chunk are CARRIED from the previous chunk (one register per thread and layer) into the tile's top pad rows -- the
    // left halo is not recomputed: base ES keeps 110 of 128 rows instead of 92.  Without STREAM (dx2 = 128: two workgroups per CU
    // already balance the chip, and 768 = 7 x 112 leaves nothing to gain) a segment is one chunk.
    const int s0 = seg * p.seg_len, s1 = min(s0 + p.seg_len, p.L_out);
    if (s0 >= p.L_out) return;
    int f0 = 0, g0 = 0, f_lo = 0, out_hi = 0;   // this chunk: frame of tile row 0 in block 0 / in the last block, first / one-past-last frame it stores
    const int n_layers = p.n_blocks * p.block_depth;
    // every read of the packed blob is a buffer load: resource + wave-uniform byte offset in SGPRs, one lane-offset VGPR for all of
    // them (64-bit per-lane pointers into the blob, live across the layer loop, were most of the kernel's register spills)
    const DecLay<DX2, KD> lay(p.d4, p.n_blocks, p.block_depth);
    const BufRsrc brs = make_rsrc(p.blob, (long)lay.total * (long)sizeof(float));
    unsigned tid16 = (unsigned)tid * 16u, lane16 = (unsigned)lane * 16u;
    auto blob_ld = [&](int float_off, unsigned voff) __attribute__((always_inline)) { return buf_ld4s(brs, voff, (unsigned)(float_off * 4)); };
#ifdef ESMI_DEC_TRACE
    int tr_n = 0;
    const bool tr_on = p.trace && seg == (DX2 > 128 ? 0 : 3) && b == p.B / 2 + 5 && lane == 0;
#define ESMI_STAMP() do { if (tr_on && tr_n < 59) p.trace[w * 64 + tr_n] = (long long)__builtin_amdgcn_s_memtime(); ++tr_n; } while (0)
    // stamps of the traced workgroup (tools/dec_budget.py): 0 = entry, 1 = sources / parameters staged, 2 = first stage (h0 gather or proj) done,
    // 3 + 12 l + (0..11) = layer l (see the layer loop; the first four layers fit); fixed slots of the FIRST chunk: 59 = layers done, 60 = mel
    // K loop issued, 61 = rows stored; slots 62 / 63 = the 100 MHz clock at entry / at the end of the first chunk (the shader clock the
    // workgroup ran at = stamp span / that span)
#define ESMI_STAMP_AT(slot) do { if (tr_on && ck == 0) p.trace[w * 64 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    if (tr_on) p.trace[w * 64 + 62] = clock_real100();
    ESMI_STAMP();
#else
#define ESMI_STAMP() do {} while (0)
#endif

    // ---- parameter staging by LDS-DMA (global_load_lds_dwordx4: memory -> LDS, 16 bytes per lane, no staging registers): issued
    // when the last reader of the slots' old contents has passed a barrier, drained by the barrier in front of the first reader of
    // the new ones.  "layer" n_layers is the mel Linear (its bias goes to the group A slots, unused by then).
    auto stage = [&](int float_off, float* dst, int n4) __attribute__((always_inline)) {   // n4 float4 from blob + float_off to dst, thread tid -> dst + 4 tid
        // (opaque: the per-lane source pointer is formed here, not hoisted out of the layer loop as a live 64-bit register pair)
        if (tid < n4) lds_dma16(p.blob + float_off + opaque_i(4 * tid), dst + 256 * w, lane);
    };
    auto fetch_A = [&](int l) __attribute__((always_inline)) {
        if (l < n_layers) stage(lay.layer0 + l * lay.layer_stride, pbuf, NA4);
    };
    auto fetch_B = [&](int l) __attribute__((always_inline)) {
        if (l < n_layers) {
            stage(lay.layer0 + l * lay.layer_stride + lay.l_pwb, pbuf + P_PWB, NB4);
            if (((l + 1) % p.block_depth) == 0) {   // block end: skip LN params, threads NB4 .. NB4 + DX2/2 (whole waves: NB4 is a multiple of 64 floats4? no -- see below)
                if (tid >= NB4 && tid < NB4 + DX2 / 2)
                    lds_dma16(p.blob + (lay.skip0 + (l / p.block_depth) * 2 * DX2) + opaque_i(4 * (tid - NB4)), pbuf + P_PWB + 256 * w, lane);
            }
        } else {
            stage(lay.mel_b, pbuf, DX2 / 4);
        }
    };
    auto commit_A = [&](int) __attribute__((always_inline)) {};
    auto commit_B = [&](int) __attribute__((always_inline)) {};

    constexpr bool STREAM = DX2 > 128;
    // STREAM: the k/2 input rows of every conv layer in front of the next chunk are carried: in LDS behind the tile (`carry_lds_layers`
    // slots), else in a global scratch row set of this workgroup (written and read back by the same thread, one chunk apart: no fence)
    float* const cws = STREAM && p.carry_ws ? p.carry_ws + ((long)seg * p.B + b) * p.ws_stride : nullptr;
    constexpr bool LNA = ESMI_DEC_LN_ACC && DX2 > 128;                // LayerNorm on the accumulators (see ln_acc below)
    float* const stat = reinterpret_cast<float*>(src + kDecRows);     // LNA: [2 exchanges][kDecRows][NS] x (sum, sum of squares)
    float* const cbuf = stat + (LNA ? 2 * kDecRows * NS * 2 : 0);     // [carry_lds_layers][PAD][DX2]
    float cnext = 0.0f;                      // the carried element of the NEXT conv layer, requested one phase ahead
    // BLOCK SKEW (round 6).  Inside a block the tile's rows keep their frames (the skip tensor lives in the row owners' registers), so
    // every conv layer costs k/2 valid rows on the right: `sh` = block_depth * k/2 per block.  At a block boundary nothing is held in
    // registers across it except that skip tensor -- which IS the tile at that moment -- so the block-end LayerNorm writes its rows `sh`
    // rows further down, the top `sh` rows (and the next conv layer's k/2 pad rows) come from the previous chunk's same boundary (the
    // block carry, scratch), and the owners re-read their skip rows: the next block starts with 128 valid rows again, `sh` frames
    // earlier.  A chunk therefore advances by keep = 128 - sh frames (base ES 122, small 124) instead of 128 - halo (110 / 116);
    // tile row r of block b holds frame g0 + (n_blocks - 1 - b) * sh + r.
    const bool skew = STREAM && cws && p.skew;
    const int sh = skew ? PAD * p.block_depth : 0;
    const int off_last = sh * (p.n_blocks - 1);
    const int bc_n4 = (sh + PAD) * CG;                                // float4 per block carry (<= kDecBlockCarry4: one per thread)
    float* const bcw = skew ? cws + n_layers * (PAD * DX2) : nullptr; // [block boundary][kDecBlockCarry4] float4
    // rows in front of the segment's first output frame: none at the start of an utterance (frames < 0 are zero rows); a segment that
    // starts inside an utterance has nothing carried in and recomputes what its first chunk lacks
    const int hl0 = s0 > 0 ? (skew ? 2 * p.halo - sh : p.halo) : off_last;
    const int keep = skew ? kDecRows - sh : kDecRows - p.halo;   // tile rows of a chunk that stay valid through every layer
    bool edge_window = false;
    int fb_cur = 0;                          // frame of tile row 0 in the current block (wave-uniform)
    // measurement aid (g_dec_clk above): the FIRST workgroup stamps {shader clock, 100 MHz clock} when it starts -> slot 0, and again
    // -> slot 1 at the start of each later chunk (dx2 = 256) / in front of its mel stage (dx2 = 128, one chunk).  (slot 1 - slot 0) is a
    // long stretch of the workgroup's life: shader ticks / 100 MHz ticks = the clock the CU ran at.  Both stamps come from ONE workgroup:
    // the s_memtime counters of different CUs are not comparable (a first version differenced two workgroups and read 9 GHz on some
    // boxes).  No stamp at the very end of the kernel (it costs a spilled register) nor inside the one-chunk kernel's chunk loop (the
    // store un-hoists the loop's address arithmetic: 28 spills).
    auto clk_stamp = [&](int ck_) __attribute__((always_inline)) {
        long long* const clk = g_dec_clk;
        if (clk && threadIdx.x == 0) {
            if (blockIdx.x == 0) {
                const int slot = ck_ == 0 ? 0 : 1;
                clk[2 * slot] = clock_shader();
                clk[2 * slot + 1] = clock_real100();
            }
        }
    };
    if constexpr (!STREAM) clk_stamp(0);
    for (int ck = 0;; ++ck) {
    if constexpr (STREAM) {
        // the thread indices pass through an opaque move per chunk: otherwise everything derived from them is loop-invariant, LICM
        // hoists it all out of the chunk loop and the kernel spills (46 VGPRs measured)
        tid = opaque_i(tid);
        lane = tid & 63; i = lane & 31; h = lane >> 5;
        tid16 = (unsigned)tid * 16u; lane16 = (unsigned)lane * 16u;
    }
    if constexpr (STREAM) clk_stamp(ck);
    g0 = s0 - hl0 + ck * keep;
    f0 = g0 + off_last;
    f_lo = max(g0, s0);
    if ((!(STREAM && cws) && ck > 0) || f_lo >= s1) break;
    out_hi = min(g0 + keep, s1);
    if (f_lo >= valid_end) {  // the rest of the segment is padding: the final masked_fill (or the [L, L_out) tail) zeroes it
        const int n = (s1 - f_lo) * p.n_mel;
        float* o = p.mel + ((long)b * p.L_out + f_lo) * p.n_mel;
        for (int e = tid; e < n; e += kDecThreads) o[e] = 0.0f;
        break;
    }
    if (ck > 0) __syncthreads();             // the previous chunk's mel stage is done with the parameter slots and the tile
    // STREAM: conv layer l's PAD input rows in front of this chunk (saved by the previous chunk) are requested with `carry_load`
    // a phase ahead and written to the tile's top pad rows by `carry_put` at the end of the phase in front of the layer's
    // depthwise conv; `carry_save` keeps this chunk's rows [keep - PAD, keep) of the same tensor for the next chunk.
    // (`slot`: the conv layer's carry slot; -1 = the first layer of a later block under the block skew, whose pad rows arrive with the
    // block carry)
    auto carry_load = [&](int slot) __attribute__((always_inline)) {
        if constexpr (STREAM) {
            cnext = 0.0f;
            if (slot >= 0 && cws && ck > 0 && tid < PAD * DX2) {
                if (slot < p.carry_lds_layers) cnext = cbuf[slot * (PAD * DX2) + tid];
                else cnext = cws[slot * (PAD * DX2) + opaque_i(tid)];   // (opaque: the address is not kept live between the phases)
            }
        }
    };
    auto carry_put = [&](int slot) __attribute__((always_inline)) {
        if constexpr (STREAM) {
            if (slot >= 0 && tid < PAD * DX2) xs[(kDecPadRows - PAD + tid / DX2) * LDSROW + tid % DX2] = cnext;
        }
    };
    auto carry_save = [&](int slot) __attribute__((always_inline)) {
        if constexpr (STREAM) {
            if (slot >= 0 && cws && tid < PAD * DX2) {
                const float v = xs[(kDecPadRows + keep - PAD + tid / DX2) * LDSROW + tid % DX2];
                if (slot < p.carry_lds_layers) cbuf[slot * (PAD * DX2) + tid] = v;
                else cws[slot * (PAD * DX2) + opaque_i(tid)] = v;
            }
        }
    };
    // The chunk's rows -> their phonemes (the length regulator's gather, networks.py:233-244).  The search used to run in global memory:
    // log2(T) DEPENDENT L2 round trips by two of the eight waves before anything else could start -- with the gather and the first
    // LayerNorm behind it the prologue was 12.8 % of a dx2 = 128 workgroup's life (profiles/r06_dec_budget.md).  Now the utterance's scan
    // row is copied into the (still unused) tile by all threads, next to the other loads of the prologue, and searched there.
    int* const cum_s = reinterpret_cast<int*>(xs + kDecPadRows * LDSROW);
    const bool cum_lds = ESMI_DEC_CUM_LDS && p.cum && p.T <= kDecRows * LDSROW;
    if (cum_lds) {
        const int* crow = p.cum + b * p.T;
        for (int e = tid; e < p.T; e += kDecThreads) cum_s[e] = crow[e];
    }
    for (int e = tid; e < 2 * kDecPadRows * LDSROW; e += kDecThreads) {
        const int r = e / LDSROW, c = e - r * LDSROW;
        const int rr = r < kDecPadRows ? r : kDecRows + r;             // rows 0,1 and 130,131
        xs[rr * LDSROW + c] = 0.0f;
    }
    if (tid < NB4)                                                     // proj_b, proj_g, proj_beta -> group B
        reinterpret_cast<f32x4*>(pbuf + P_PWB)[tid] = blob_ld(lay.proj_b, tid16);
    carry_load(0);
    fetch_A(0);
    commit_A(0);
#ifdef ESMI_DEC_TRACE
    ESMI_STAMP_AT(52);   // prologue loads issued
#endif
    if (cum_lds) __syncthreads();
#ifdef ESMI_DEC_TRACE
    ESMI_STAMP_AT(53);   // scan row in LDS
#endif
    if (tid < kDecRows) {
        const int f = f0 + tid;
        int s;
        if (f < 0 || f >= L) s = -1;                                   // outside the padded sequence
        else if (p.cum) {
            if (f < mlen) {
                int ph;
                if (cum_lds) {                                          // first i with cum[i] > f (searchsorted right); T if none
                    int lo = 0, hi = p.T;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (cum_s[mid] > f) hi = mid;
                        else lo = mid + 1;
                    }
                    ph = lo;
                } else ph = frame_to_phoneme(p.cum + b * p.T, p.T, f);
                s = ph < p.T ? b * p.T + ph : -2;
            } else s = -2;                                             // padding frame: zero input row
        } else s = b * L + f;
        src[tid] = s;
    }
    __syncthreads();
    ESMI_STAMP();   // 1: sources / parameters staged

    // LayerNorm ownership: lane = 16*rg + c; the thread owns rows ln_row0 + (0..3) and the float4 channel groups c + 16*k of
    // each.  The four row groups of a wave sit 16 rows apart (rows 64(w>>2) + 16rg + 4(w&3) + j): the LDS bank of a lane's
    // 16-byte access is 4*(row + c) mod 64, so rows that agree mod 16 make every ds_read_b128 / ds_write_b128 of the pass
    // conflict-free; in general rows 64(w / LNPER) + 16rg + RPT(w % LNPER) + j (with four CONSECUTIVE row quadruples per wave, lanes of neighbouring groups met in the same banks:
    // 17.8 % of the kernel's LDS cycles were bank conflicts, profiles/r01_l).
    constexpr int LNPER = 16 / RPT;         // waves that share one residue class of rows mod 16
    int ln_c = lane & (TPR - 1), ln_row0 = 64 * (w / LNPER) + 16 * (lane / TPR) + RPT * (w % LNPER);
    // rows of the tile that lie outside [0, L) for the block whose tile row 0 holds frame `fbase`
    auto set_edge = [&](int fbase) __attribute__((always_inline)) {
        edge_window = fbase < 0 || fbase + kDecRows > L;   // some window rows lie outside [0, L) (SGPR: a scalar branch)
        fb_cur = fbase;                          // row r exists in the reference iff 0 <= fb_cur + r < L (`row_inside`: derived where it is
                                                 // used -- a per-thread mask kept across the layer loop is a register the dx2 = 256 kernel spills)
    };
    auto row_inside = [&](int j) __attribute__((always_inline)) { return (unsigned)(fb_cur + ln_row0 + j) < (unsigned)L; };
    set_edge(f0);
    f32x16 acc[MT][NTW];
    f32x16 skipA[LNA ? MT : 1][LNA ? NTW : 1];   // LNA: the block's skip tensor in the accumulators' layout
    f32x4 skip[RPT][NV];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
#pragma unroll
        for (int v = 0; v < NV; ++v) skip[j][v] = zero4();
    }
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
    constexpr int KSUB = DX2 <= 128 ? (SPLIT ? 4 : 8) / NTW : (NW > 8 ? 2 : 8);   // k-steps (of 8 channels) of weights in registers at a time
#if ESMI_DEC_SPLIT
    constexpr int KS16 = KSUB / 2;
    u32x4 bf[NTW][KS16][2];
    auto load_b = [&](int wsl, int k0) __attribute__((always_inline)) {   // wsl: float offset of the wave's weight slice in the blob
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
    auto wslice = [&](int off, int c) __attribute__((always_inline)) { return off + (c * (DX2 / 32) + ns * NTW) * 8 * 2 * 256; };

    // The A operand rows already stored as the two f16 planes (row = [DX2 halves h1 | DX2 halves h2 | pad], written by the
    // depthwise phase / the last LayerNorm): per (16-channel step, row tile) two ds_read_b128 + 3*NTW MFMAs.
    // Two forms.  WD == 0 (dx2 = 128, two workgroups per CU): the weights of KSUB k-steps are loaded, then used -- the compiler
    // keeps one or two fragments in flight and the neighbour workgroup's VALU phases fill the L2 round trips (a ring that
    // pipelines the loop measured slower there: profiles/r03_probes/decoder_round3_experiments.md).  WD > 0 (dx2 = 256, ONE workgroup
    // per CU, nobody to fill the gaps): hand-pipelined -- an item = (16-channel step s, row tile mt); weight fragments of step s + WD
    // and A fragments of item q + AD are requested while item q's MFMAs run (VGPR rings; scheduling fences keep hipcc from sinking
    // the loads back to their first use).  small ES decoder 1774 -> 1687 us with WD = 2; WD = 4 spills (1736), 6: 1976.
    constexpr int WD = DX2 > 128 ? kDecWeightRing256 : 0, AD = kDecARing;
    constexpr int NSTEP = 8 * KCH, NITEM = NSTEP * MT;
    static_assert(WD >= 0 && WD <= NSTEP && AD >= 1 && AD <= NITEM, "ring depths");
    u32x4 wr[WD > 0 ? WD : 1][NTW][2];
    // (`ntc`: 32-column tiles per wave of THIS contraction -- NTW for the conv layers; 1 for the mel Linear of the dx2 = 256 kernel, whose
    // n_mel <= 96 columns are packed as three one-tile slices so that three SIMDs share them instead of two)
    auto w_fetch = [&](int off, int s, int slot, auto ntc) __attribute__((always_inline)) {
        constexpr int NT = decltype(ntc)::value;
        const int wsl = off + ((s >> 3) * (4 * NT) + ns * NT) * 8 * 2 * 256;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wr[slot][t][pl] = __builtin_bit_cast(u32x4, blob_ld(wsl + ((t * 8 + (s & 7)) * 2 + pl) * 256, lane16));
        }
    };
    // the first WD weight steps of the matrix at `off` (issued ahead of the barrier that precedes the K loop: the L2 round
    // trip then overlaps the barrier wait)
    auto gemm_prefetch = [&](int off, auto ntc) __attribute__((always_inline)) {
        if constexpr (WD > 0) {
#pragma unroll
            for (int s = 0; s < WD; ++s) w_fetch(off, s, s, ntc);
            sched_fence();
        }
    };
    auto gemm_planes = [&](int off, auto ntc) __attribute__((always_inline)) {
        constexpr int NT = decltype(ntc)::value;
        static_assert(WD > 0 || NT == NTW, "the un-pipelined form keeps the layers' slice width");
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
                for (int t = 0; t < NT; ++t)
                    acc[mt][t] = mfma32_split2_wx(wr[s % (WD > 0 ? WD : 1)][t][0], wr[s % (WD > 0 ? WD : 1)][t][1], ar[q % AD], acc[mt][t]);
                if (q + AD < NITEM) a_fetch(q + AD, q % AD);
                if (mt == MT - 1 && s + WD < NSTEP) w_fetch(off, s + WD, s % (WD > 0 ? WD : 1), ntc);
                sched_fence();
            }
        }
    };
#else
    f32x4 bf[NTW][KSUB];
    auto load_b = [&](int wsl, int k0) __attribute__((always_inline)) {   // wsl: float offset of the wave's weight slice in the blob
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
    auto wslice = [&](int off, int c) __attribute__((always_inline)) { return off + (c * (DX2 / 32) + ns * NTW) * 16 * 256; };
    auto gemm_prefetch = [&](int, auto) __attribute__((always_inline)) {};
    auto gemm_planes = [&](int, auto) __attribute__((always_inline)) {};
#endif
    // full dx2-wide contraction over fp32 rows of the tile, un-pipelined
    auto gemm_rows = [&](int off) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += KSUB) {
                load_b(wslice(off, c), k0);
                mma_sub(128 * c, k0);
            }
        }
    };

    // ================================================================== epilogues
    // accumulators (+ bias, tanh) -> tile.  The products are computed TRANSPOSED (weights as the first MFMA operand): lane
    // (i, h) holds frame i of the tile and, per register quad g = r >> 2, the four consecutive channels 8g + 4h .. + 3 --
    // one ds_write_b128 per quad instead of four ds_write_b32 (and float4 global stores for the mel rows).
    // Two steps: `tanh_acc` turns the accumulators into tanh(acc + bias) IN PLACE -- it touches no tile row, so it runs right behind
    // the K loop, before the barrier that waits for the last reader of the operand planes: a wave that is through its MFMAs spends the
    // transcendental-heavy part of the epilogue while the other waves of its SIMD still feed the matrix pipe -- and `store_acc`
    // writes them to the tile after that barrier.
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
    auto store_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            float* base = xs + opaque_i((kDecPadRows + 32 * MT * mh + i) * LDSROW + ns * WCOLS + 32 * t + 4 * h);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[mt][t][4 * g + e];
                    *reinterpret_cast<f32x4*>(base + 32 * mt * LDSROW + 8 * g) = v;
                }
            }
        }
    };
    // ---- LNA (dx2 = 256): LayerNorm without the tile round trip.  After the K loop lane (i, h) of wave (mh, ns) holds, per row tile mt, row
    // 64 mh + 32 mt + i x 32 NTW channels of the wave's column slice.  `stats_put`: the lane's sum and sum of squares, + its partner half
    // (lane ^ 32), written as one (sum, sumsq) pair per (row, ns) -- ahead of the barrier the epilogue needs anyway (last reader of the
    // operand planes); behind it every lane adds the row's four pairs (`stats_get`: two ds_read_b128), normalises its registers
    // (`ln_acc`: gain / shift of its own channels, read like the bias in tanh_acc) and stores the finished rows once (`store_ln`).
    // One-pass variance E[x^2] - mean^2 on |x| <= 1 (tanh) or O(1) (block end) values: parity held at 1e-5 (tests).  Against the
    // round-5 form: one barrier, 16 ds_write_b128 + 16 ds_read_b128 per thread and the row owners' DPP reductions less per layer.
    auto stats_put = [&](int buf) __attribute__((always_inline)) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s1 += acc[mt][t][r];
                    s2 = fmaf(acc[mt][t][r], acc[mt][t][r], s2);
                }
            }
            s1 += swap32_f(s1);
            s2 += swap32_f(s2);
            if (h == 0) *reinterpret_cast<f32x2*>(stat + opaque_i(((buf * kDecRows + 32 * MT * mh + i) * NS + ns) * 2) + 32 * mt * NS * 2) = f32x2{s1, s2};
        }
    };
    auto stats_get = [&](int buf, float (&mean)[MT], float (&rstd)[MT]) __attribute__((always_inline)) {
        const float* sp = stat + opaque_i((buf * kDecRows + 32 * MT * mh + i) * NS * 2);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sp + 32 * mt * NS * 2), b = *reinterpret_cast<const f32x4*>(sp + 32 * mt * NS * 2 + 4);
            const float s1 = (a[0] + a[2]) + (b[0] + b[2]), s2 = (a[1] + a[3]) + (b[1] + b[3]);
            mean[mt] = s1 * (1.0f / DX2);
            rstd[mt] = ESMI_DEC_RSQRT(fmaxf(fmaf(-mean[mt], mean[mt], s2 * (1.0f / DX2)), 0.0f) + 1e-5f);
        }
    };
    auto ln_acc = [&](const float* gain, const float* shift, const float (&mean)[MT], const float (&rstd)[MT]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const float* gp = gain + opaque_i(ns * WCOLS + 32 * t + 4 * h);
            const float* bp = shift + opaque_i(ns * WCOLS + 32 * t + 4 * h);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(gp + 8 * g), bv = *reinterpret_cast<const f32x4*>(bp + 8 * g);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[mt][t][4 * g + e] = fmaf((acc[mt][t][4 * g + e] - mean[mt]) * rstd[mt], gv[e], bv[e]);
                }
            }
        }
    };
    // the finished rows -> tile: fp32 rows `shw` rows further down (block skew; rows that would leave the tile are the rows the block lost),
    // or (the last LayerNorm) the mel Linear's two f16 operand planes; rows outside [0, L) are zero rows
    auto store_ln = [&](bool planes_, int shw) __attribute__((always_inline)) {
        bool inside[MT], keep_row[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = 32 * MT * mh + 32 * mt + i;
            inside[mt] = !edge_window || (unsigned)(fb_cur + row) < (unsigned)L;
            keep_row[mt] = row + shw < kDecRows;
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            float* base = xs + opaque_i((kDecPadRows + 32 * MT * mh + i + shw) * LDSROW + ns * WCOLS + 32 * t + 4 * h);
            unsigned* pbase = reinterpret_cast<unsigned*>(xs) + opaque_i((kDecPadRows + 32 * MT * mh + i) * LDSROW + (ns * WCOLS + 32 * t + 4 * h) / 2);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = inside[mt] ? acc[mt][t][4 * g + e] : 0.0f;
                    if (SPLIT && planes_) {
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        unsigned h1a, h2a, h1b, h2b;
                        split_f16_pair(v[0], v[1], h1a, h2a);
                        split_f16_pair(v[2], v[3], h1b, h2b);
                        unsigned* rowp = pbase + 32 * mt * LDSROW + 4 * g;
                        *reinterpret_cast<u32x2*>(rowp) = u32x2{h1a, h1b};
                        *reinterpret_cast<u32x2*>(rowp + DX2 / 2) = u32x2{h2a, h2b};
                    } else if (keep_row[mt]) {
                        *reinterpret_cast<f32x4*>(base + 32 * mt * LDSROW + 8 * g) = v;
                    }
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
                v[k][e] -= mean;
                q = fmaf(v[k][e], v[k][e], q);
            }
        }
        q = row_sum_n<TPR>(q);
        const float rstd = ESMI_DEC_RSQRT(q * (1.0f / DX2) + 1e-5f);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[k][e] = fmaf(v[k][e] * rstd, g[k][e], be[k][e]);
        }
    };
    auto ln_params = [&](const float* pv, f32x4 (&o)[NV]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NV; ++k) o[k] = *reinterpret_cast<const f32x4*>(pv + 4 * TPR * k);
    };
    // LN pass over the tile (in place): x = LN(x) [; x = LN_s(x + skip), skip = x at a block end]; rows outside [0, L) -> 0.
    // PLANES: the rows are written as the two f16 planes the pipelined K loop reads (the last LayerNorm feeds the mel Linear
    // only), otherwise as fp32.  BLOCK_END is a template-like constant at both call sites so that `skip = v` costs nothing.
    // ONE body with workgroup-uniform branches (block end / operand planes / shifted rows): four template-like copies of the pass made
    // the register allocator shuffle the 64 skip registers at every join (340 moves per layer on dx2 = 256) and spill around them.
    // `shw` (block end, fp32 rows only): block skew -- the rows are written `shw` rows further down (those that would leave the tile are
    // dropped: they are the rows the block lost), behind a barrier that waits for every owner to have read its rows.
    auto ln_pass = [&](const float* pb0, bool block_end_, bool planes_, int shw) __attribute__((always_inline)) {
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
#pragma unroll
        for (int j = 0; j < RPT; ++j) ln_regs(v[j], g, be);
        if (block_end_) {  // end of a decoder block: skip = LN_s(x + skip), networks.py:299
            ln_params(pb + P_SG, g);
            ln_params(pb + P_SB, be);
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
#pragma unroll
                for (int k = 0; k < NV; ++k) v[j][k] += skip[j][k];
                ln_regs(v[j], g, be);
            }
        }
        if (edge_window) {   // workgroup-uniform: only the first / last windows of an utterance hold rows outside [0, L)
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
#pragma unroll
                for (int k = 0; k < NV; ++k)
                    if (!row_inside(j)) v[j][k] = zero4();
            }
        }
        if (SPLIT && planes_) {   // the last LayerNorm feeds the mel Linear: the rows leave as its two f16 operand planes
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    unsigned h1a, h2a, h1b, h2b;
                    split_f16_pair(v[j][k][0], v[j][k][1], h1a, h2a);
                    split_f16_pair(v[j][k][2], v[j][k][3], h1b, h2b);
                    unsigned* rowp = reinterpret_cast<unsigned*>(ln_ptr + j * LDSROW) - 2 * ln_c + 2 * TPR * k;   // dword 2*(channel group)
                    *reinterpret_cast<u32x2*>(rowp) = u32x2{h1a, h1b};
                    *reinterpret_cast<u32x2*>(rowp + DX2 / 2) = u32x2{h2a, h2b};
                }
            }
            return;
        }
        float* wp = ln_ptr;
        if (STREAM && shw) {   // (workgroup-uniform) block skew: every owner has read its rows before any row moves
            __syncthreads();
            wp = ln_ptr + shw * LDSROW;
        }
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            if (STREAM && ln_row0 + j + shw >= kDecRows) continue;   // (rows the block lost; never without the skew)
#pragma unroll
            for (int k = 0; k < NV; ++k) *reinterpret_cast<f32x4*>(wp + j * LDSROW + 4 * TPR * k) = v[j][k];
        }
        // (a block end does not keep its result as the new skip tensor in registers: the owners re-read their rows from the tile when
        // the next block starts)
    };
    typedef std::true_type TrueC;
    typedef std::false_type FalseC;
    // 32-column tiles per wave: conv layers; mel Linear (split build of the dx2 = 256 kernel: one, on three column slices)
    constexpr int NTM = (SPLIT && DX2 > 128) ? 1 : NTW;
    typedef std::integral_constant<int, NTW> NtwC;
    typedef std::integral_constant<int, NTM> NtmC;

    // ---- proj: Linear(d4, dx2) + Tanh + LN.  All three are row-wise, and a frame's input row is its phoneme's row: when
    // the caller supplies h0 = LN(tanh(proj(x))) at PHONEME rate (enc_fuse_va_kernel computes it while the features
    // are still on the CU) the stage reduces to a gather -- one of the six GEMM stages of the window disappears
    // (D frames per phoneme share one row).  Padding frames (zero input rows) get LN(tanh(proj_b)).
    if (p.h0) {
        static_assert(kDecRows * (DX2 / 4) % kDecThreads == 0, "whole passes");
        for (int it = 0; it < kDecRows * (DX2 / 4) / kDecThreads; ++it) {   // (scalar trip counter: a per-lane one is a 64-bit register pair the dx2 = 256 kernel spilled)
            const int e = tid + it * kDecThreads;
            const int r = e / (DX2 / 4), q = e - r * (DX2 / 4);
            const int s = src[r];
            f32x4 v = zero4();
#if ESMI_DEC_H0_NT
            if (s >= 0) v = STREAM ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.h0 + (long)s * DX2 + 4 * q)) : ld4(p.h0 + (long)s * DX2 + 4 * q);
#else
            if (s >= 0) v = ld4(p.h0 + (long)s * DX2 + 4 * q);
#endif
            else if (s == -2) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(pbuf + P_PWB + 4 * q);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = ESMI_DEC_TANH(bb[c]);
            }
            *reinterpret_cast<f32x4*>(xs + (kDecPadRows + r) * LDSROW + 4 * q) = v;
        }
#ifdef ESMI_DEC_TRACE
        ESMI_STAMP_AT(54);   // h0 rows requested and written to the tile
#endif
        __syncthreads();
#ifdef ESMI_DEC_TRACE
        ESMI_STAMP_AT(55);   // barrier
#endif
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
        carry_put(0);
        __syncthreads();
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
                load_b(wslice(0, ch), k0);
                mma_sub(0, k0);
            }
        }
        tanh_acc(pbuf + P_PWB);
        __syncthreads();  // every wave finished reading the staged input
        store_acc();
        __syncthreads();
        {   // LN(tanh(proj)); skip = the stage's output
            const float* pb = pbuf + opaque_i(4 * ln_c);
            float* ln_ptr = xs + opaque_i((kDecPadRows + ln_row0) * LDSROW + 4 * ln_c);
            f32x4 g[NV], be[NV];
            ln_params(pb + P_G, g);
            ln_params(pb + P_B, be);
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
#pragma unroll
                for (int k = 0; k < NV; ++k) skip[j][k] = *reinterpret_cast<const f32x4*>(ln_ptr + j * LDSROW + 4 * TPR * k);
                ln_regs(skip[j], g, be);
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    if (!row_inside(j)) skip[j][k] = zero4();
                    *reinterpret_cast<f32x4*>(ln_ptr + j * LDSROW + 4 * TPR * k) = skip[j][k];
                }
            }
        }
        carry_put(0);
        __syncthreads();
    }

    ESMI_STAMP();   // 2: first stage done
    // ---- conv layers
    int dw_cg = tid % CG, dw_r0 = (tid / CG) * RS;
    int blk = 0, lin = 0;                    // block of layer l and the layer's place in it
    for (int l = 0; l < n_layers; ++l) {
        const float* pb = pbuf;
        const int lbase = lay.layer0 + l * lay.layer_stride;
        const bool block_end = lin + 1 == p.block_depth;
        // conv-layer carry slots (see carry_load): without the skew one per layer; with it the first layer of blocks >= 1 has none
        const int slot = skew ? ((lin == 0 && l > 0) ? -1 : l - blk) : l;
        const int slot_next = skew ? (block_end ? -1 : l + 1 - blk) : l + 1;
        if constexpr (STREAM) {   // (as at the top of the chunk loop: keeps the layer's address arithmetic from being hoisted and spilled)
            tid = opaque_i(tid);
            lane = tid & 63; i = lane & 31; h = lane >> 5;
            tid16 = (unsigned)tid * 16u; lane16 = (unsigned)lane * 16u;
            ln_c = lane & (TPR - 1); ln_row0 = 64 * (w / LNPER) + 16 * (lane / TPR) + RPT * (w % LNPER);
            dw_cg = tid % CG; dw_r0 = (tid / CG) * RS;
            if (skew && lin == 0 && l > 0) {
                // a block starts on the shifted tile (complete since the barrier behind the last LayerNorm): hand the rows in front of
                // the NEXT chunk's tile to the block carry (tile rows [keep - PAD, 128): the same thread reads them back one chunk
                // later) and derive which rows lie outside [0, L) at this block's frame base
                if (tid < bc_n4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(xs + (kDecPadRows + keep - PAD + tid / CG) * LDSROW + 4 * (tid % CG));
                    *reinterpret_cast<f32x4*>(bcw + (blk - 1) * (4 * kDecBlockCarry4) + opaque_i(4 * tid)) = v;
                }
                set_edge(g0 + sh * (p.n_blocks - 1 - blk));
            }
        }
        if constexpr (LNA) {
            if (lin == 0) {        // a block starts (l = 0: behind the first stage): the tile is the skip tensor, read in the accumulators' layout
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    const float* base = xs + opaque_i((kDecPadRows + 32 * MT * mh + i) * LDSROW + ns * WCOLS + 32 * t + 4 * h);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(base + 32 * mt * LDSROW + 8 * g);
#pragma unroll
                            for (int e = 0; e < 4; ++e) skipA[mt][t][4 * g + e] = v[e];
                        }
                    }
                }
            }
        }
        if (!LNA && lin == 0 && l > 0) {   // a block starts: its input (the tile, complete since the barrier behind the last LayerNorm) is the skip tensor
            const float* ln_ptr = xs + opaque_i((kDecPadRows + ln_row0) * LDSROW + 4 * ln_c);
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
#pragma unroll
                for (int k = 0; k < NV; ++k) skip[j][k] = *reinterpret_cast<const f32x4*>(ln_ptr + j * LDSROW + 4 * TPR * k);
            }
        }
        ESMI_STAMP();   // 0: layer start
        // 1. depthwise conv in place: window -> registers | barrier | filtered rows -> tile (as the K loop's operand planes)
        {
            f32x4 win[RS + 2 * PAD];
            float* col = xs + opaque_i((kDecPadRows + dw_r0 - PAD) * LDSROW + 4 * dw_cg);
            const float* pbt = pb + opaque_i(4 * dw_cg);
            unsigned* prow = reinterpret_cast<unsigned*>(xs) + opaque_i((kDecPadRows + dw_r0) * LDSROW + 2 * dw_cg);
            fetch_B(l);          // this layer's bias / LN params: global -> register now, -> LDS at the end of the phase
#pragma unroll
            for (int r = 0; r < RS + 2 * PAD; ++r) win[r] = *reinterpret_cast<const f32x4*>(col + r * LDSROW);
            f32x4 tap[KD];       // in registers for all RS rows
#pragma unroll
            for (int j = 0; j < KD; ++j) tap[j] = *reinterpret_cast<const f32x4*>(pbt + j * DX2);
            const f32x4 tb = *reinterpret_cast<const f32x4*>(pbt + P_DWB);
            carry_save(slot);    // (this layer's input rows in front of the NEXT chunk, before the planes overwrite them)
            ESMI_STAMP();   // 1: window loaded (issued)
            __syncthreads();
            ESMI_STAMP();   // 2: barrier passed
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                f32x4 a = tb;
#pragma unroll
                for (int j = 0; j < KD; ++j) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = fmaf(win[r + j][e], tap[j][e], a[e]);
                }
                if (SPLIT) {   // the K loop's A operand, already split (esmi_dev.h): 4 channels = 2 dwords per plane
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
        commit_B(l);
        gemm_prefetch(lbase + lay.l_pw, NtwC{});   // first weight steps: in flight across the barrier
        ESMI_STAMP();   // 3: dw written
        __syncthreads();
        ESMI_STAMP();   // 4: barrier
        // 2. pointwise conv: K = dx2
        zero_acc();
        if (SPLIT) gemm_planes(lbase + lay.l_pw, NtwC{});
        else gemm_rows(lbase + lay.l_pw);
        ESMI_STAMP();   // 5: K loop issued
        // 3. bias + tanh on the accumulators (no tile access: ahead of the barrier), then -> tile
        tanh_acc(pb + P_PWB);
        if constexpr (LNA) stats_put(0);
        fetch_A(l + 1);      // next layer's taps (their slots were last read by this layer's depthwise phase)
        ESMI_STAMP();   // 6: bias + tanh done on the accumulators
        __syncthreads();  // all reads of the filtered tile done
        ESMI_STAMP();   // 7: barrier
        if constexpr (!LNA) store_acc();
        commit_A(l + 1);     // (the taps' LDS slots were last read by this layer's depthwise phase)
        ESMI_STAMP();   // 8: tanh stored
        if constexpr (!LNA) __syncthreads();
        ESMI_STAMP();   // 9: barrier
        // 4. LayerNorm (+ block-end skip LayerNorm) by row owners; the last one writes the mel Linear's operand planes
        if (l + 1 == n_layers) fetch_B(n_layers);   // mel bias -> group A slots (taps: last read by this layer's depthwise phase)
        else carry_load(slot_next);
        // block skew, at the end of every block but the last: the rows go `sh` rows down, the rows in front of them arrive from the
        // previous chunk (requested here, a phase ahead; zero rows at the start of an utterance -- frames < 0)
        const int shw = (STREAM && skew && block_end && l + 1 < n_layers) ? sh : 0;
        if constexpr (LNA) {
            float mean[MT], rstd[MT];
            stats_get(0, mean, rstd);
            ln_acc(pb + P_G, pb + P_B, mean, rstd);
            if (block_end) {   // end of a decoder block: LN_s(x + skip), networks.py:299 (a second exchange: one more barrier, as before)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int t = 0; t < NTW; ++t) acc[mt][t] += skipA[mt][t];
                }
                stats_put(1);
                __syncthreads();
                stats_get(1, mean, rstd);
                ln_acc(pb + P_SG, pb + P_SB, mean, rstd);
            }
            store_ln(SPLIT && l + 1 == n_layers, shw);
        } else {
            ln_pass(pb, block_end, SPLIT && l + 1 == n_layers, shw);
        }
        ESMI_STAMP();   // 10: LN done
        if (l + 1 == n_layers) gemm_prefetch(lay.mel_w, NtmC{});   // (the mel bias went to the unused group A slots by LDS-DMA above)
        else carry_put(slot_next);
        if constexpr (STREAM) {
            if (shw && tid < bc_n4) {  // pad rows of the next conv layer + tile rows [0, sh) of the next block
                f32x4 bcv = zero4();
                if (ck > 0) bcv = ld4(bcw + blk * (4 * kDecBlockCarry4) + opaque_i(4 * tid));
                *reinterpret_cast<f32x4*>(xs + (kDecPadRows - PAD + tid / CG) * LDSROW + 4 * (tid % CG)) = bcv;
            }
        }
        __syncthreads();
        ESMI_STAMP();   // 11: barrier
        if (++lin == p.block_depth) { lin = 0; ++blk; }
    }

#ifdef ESMI_DEC_TRACE
    ESMI_STAMP_AT(59);   // conv layers done
#endif
    // ---- mel Linear(dx2, n_mel) on skip (held in the tile), masked store
    if constexpr (!STREAM) {   // clock probe, second stamp of the one-chunk kernel: the first workgroup again, in front of its last stage
        long long* const clk = g_dec_clk;     // (s_memtime counters of different CUs are not comparable: both stamps come from one workgroup)
        if (clk && blockIdx.x == 0 && w == 0) {   // one wave stores (its 64 lanes write the same 16 bytes: the wave id is a scalar register,
            typedef long long i64x2 __attribute__((ext_vector_type(2)));   // a test for one THREAD would keep a lane register alive across the layers)
            *reinterpret_cast<i64x2*>(clk + 2) = i64x2{clock_shader(), clock_real100()};
        }
    }
    if (n_layers == 0) {   // (degenerate: proj output straight into the mel Linear, fp32 rows)
        fetch_B(0);
        __syncthreads();
    }
    const bool mel_wave = ns * 32 * NTM < p.n_mel;   // wave-uniform: column slices beyond n_mel have nothing to do
    const float* mb = pbuf;                          // mel bias (zero padded to dx2)
    const bool vec_ok = (p.n_mel & 3) == 0;          // rows of 16-byte multiples: float4 stores
    if (mel_wave) {
        zero_acc();
        if (SPLIT && n_layers > 0) gemm_planes(lay.mel_w, NtmC{});
        else gemm_rows(lay.mel_w);
    }
#ifdef ESMI_DEC_TRACE
    ESMI_STAMP_AT(60);   // mel K loop issued
#endif
    if constexpr (STREAM) {
        // dx2 = 256: the chunk's mel rows are ONE contiguous run of the output (row stride = n_mel floats), so they are put together in
        // LDS (the tile is free once every wave is through the K loop) and leave as whole-line streaming stores.  Stored straight from the
        // accumulators a wave instruction scatters 64 x 16 B over 32 rows: as plain stores those pass through the XCD's L2 and evict the
        // weight slices that every chunk re-reads (round 5: 2.95x the algorithmic HBM traffic on base ES), as streaming stores they
        // reach HBM as partial lines (measured: writes 2.8x the mel).
        float* stg = xs;                             // [kDecRows][n_mel rounded up to a multiple of 4]
        const int mstride = (p.n_mel + 3) & ~3;
        __syncthreads();
        if (mel_wave) {
#pragma unroll
            for (int t = 0; t < NTM; ++t) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int r = 32 * MT * mh + 32 * mt + i;
                    const bool live = g0 + r < valid_end;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = ns * 32 * NTM + 32 * t + 8 * g + 4 * h;
                        if (col >= p.n_mel) continue;
                        const f32x4 bc = *reinterpret_cast<const f32x4*>(mb + col);
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = live ? fmaf(acc[mt][t][4 * g + e], WSI, bc[e]) : 0.0f;
                        *reinterpret_cast<f32x4*>(stg + r * mstride + col) = v;
                    }
                }
            }
        }
        __syncthreads();
        const float* sp = stg + (f_lo - g0) * mstride;
        float* dp = p.mel + ((long)b * p.L_out + f_lo) * p.n_mel;
        if (vec_ok) {
            const int n4 = (out_hi - f_lo) * (p.n_mel >> 2);
            for (int e = tid; e < n4; e += kDecThreads) {
#if ESMI_DEC_MEL_NT
                __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(sp + 4 * e), reinterpret_cast<f32x4*>(dp + 4 * e));
#else
                *reinterpret_cast<f32x4*>(dp + 4 * e) = *reinterpret_cast<const f32x4*>(sp + 4 * e);
#endif
            }
        } else {   // rows that are not 16-byte multiples: element by element
            const int n = (out_hi - f_lo) * p.n_mel;
            for (int e = tid; e < n; e += kDecThreads) dp[e] = sp[(e / p.n_mel) * mstride + e % p.n_mel];
        }
    } else if (mel_wave) {
#pragma unroll
        for (int t = 0; t < NTM; ++t) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int f = g0 + 32 * MT * mh + 32 * mt + i;
                if (f < f_lo || f >= out_hi) continue;
                float* orow = p.mel + ((long)b * p.L_out + f) * p.n_mel;
                const bool live = f < valid_end;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = ns * 32 * NTM + 32 * t + 8 * g + 4 * h;
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
#ifdef ESMI_DEC_TRACE
    ESMI_STAMP_AT(61);   // rows stored (issued)
    if (tr_on && ck == 0) p.trace[w * 64 + 63] = clock_real100();
#endif
    }   // chunks of the segment
}

}  // namespace esmi
