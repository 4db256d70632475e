// Fully fused mel decoder -- MelDecoder.forward, layers/networks.py:291-304, plus the length
// regulator gather (networks.py:233-244) on its input side and Phoneme2Mel's final masked_fill
// (networks.py:424-427) on its output side.
//
//   skip = LN(tanh(Linear(d4,dx2)(x)))
//   n_blocks x { x = skip; block_depth x [ x = LN(tanh(Conv1x1(dwConv_k(x)))) ]; skip = LN_s(x + skip) }
//   mel  = Linear(dx2, n_mel)(skip)
//
// One 256-thread workgroup owns a 128-frame window of one utterance: TL = 128 - 2*halo frames
// are kept, halo = (k/2)*n_blocks*block_depth frames per side are recomputed so that no
// activation ever leaves the CU.  Activations live in LDS as a [132][DX2+4] fp32 tile (two zero
// rows per side give the depthwise conv its in-tile padding; +4 floats per row make the
// per-row 16-byte fragment reads bank-conflict free).  Each wave owns 32 complete rows, so
// LayerNorm statistics are wave-local (5 xor-shuffles per row in the MFMA C/D layout) and the
// skip tensor stays in registers.  Every contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32):
//   A fragment = lane (i = lane&31, h = lane>>5) reads 4 channels [8kc+4h, +4) of row i from LDS,
//                applying the depthwise k-tap filter on the fly (VALU, hidden under the MFMAs);
//   B fragment = one coalesced 16-byte load per lane from the pre-packed weight blob (L2-resident).
//
// Fidelity notes (SURVEY.md §7 "hard parts"):
//  * frames in [mel_len[b], L) are PADDING FRAMES: their input rows are zero but they are computed
//    like any other frame, because the reference computes them and the k-tap conv leaks them into
//    the last valid frames;
//  * frames outside [0, L) do not exist in the reference: every layer's Conv1d zero-pads there, so
//    such rows are forced to 0 after every LayerNorm;
//  * rows >= mel_len[b] of the output are zeroed only at the very end (the final masked_fill).
#pragma once
#include "esmi_dev.h"
#include "small_kernels.h"

namespace esmi {

constexpr int kDecRows = 128;     // frames per workgroup window
constexpr int kDecPadRows = 2;    // zero rows above/below the window in LDS (>= k/2)
constexpr int kMelNT = 3;         // mel Linear: n_mel <= 96 columns

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
    L.proj_w = o; o += (long)d4 * dx2;
    L.proj_b = o; o += dx2;
    L.proj_g = o; o += dx2;
    L.proj_beta = o; o += dx2;
    L.l_dw = 0;
    L.l_dwb = (long)kd * dx2;
    L.l_pw = L.l_dwb + dx2;
    L.l_pwb = L.l_pw + (long)dx2 * dx2;
    L.l_g = L.l_pwb + dx2;
    L.l_b = L.l_g + dx2;
    L.layer_stride = L.l_b + dx2;
    L.layer0 = o; o += L.layer_stride * n_blocks * block_depth;
    L.skip0 = o; o += 2L * dx2 * n_blocks;
    L.mel_w = o; o += (long)dx2 * 32 * kMelNT;
    L.mel_b = o; o += 32 * kMelNT;
    L.total = o;
    return L;
}

struct MelDecP {
    const float* blob;
    DecLayout lay;
    int d4, n_blocks, block_depth, n_mel;
    const float* x;        // (B,T,d4) phoneme-rate (cum != NULL) or (B,L,d4) frame-rate
    const int* cum;        // (B,T) inclusive duration cumsum or NULL
    const int* mel_len;    // (B) or NULL
    const int* lmax_dev;   // device scalar or NULL
    int lmax_host;
    int apply_mask;
    int B, T, L_out;
    float* mel;            // (B, L_out, n_mel)
    int halo, TL;
};

template <int DX2>
__host__ __device__ constexpr int dec_lds_floats(int kd) {
    return (kDecRows + 2 * kDecPadRows) * (DX2 + 4) + (kd + 1) * DX2 + kDecRows;
}

// bias + tanh + LN (+ skip add + LN) + zero rows outside [0,L) + write the tile back to LDS
template <int NT>
__device__ __forceinline__ void dec_epilogue(f32x16 (&acc)[NT], f32x16 (&skip)[NT], const float* __restrict__ bias,
                                             const float* __restrict__ g, const float* __restrict__ be,
                                             const float* __restrict__ sg, const float* __restrict__ sb, bool set_skip,
                                             float* __restrict__ xs_w, const int* __restrict__ src_w, int lane) {
    constexpr int LDSROW = 32 * NT + 4;
    const int i = lane & 31;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float bc = bias[32 * nt + i];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = tanh_f32(acc[nt][r] + bc);
    }
    layernorm_tile<NT>(acc, g, be, lane);
    if (sg) {  // end of a decoder block: skip = LN_s(x + skip), networks.py:299
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] += skip[nt];
        layernorm_tile<NT>(acc, sg, sb, lane);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = tile_row(r, lane);
        const bool inside = src_w[row] != -1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float v = inside ? acc[nt][r] : 0.0f;
            acc[nt][r] = v;
            xs_w[row * LDSROW + 32 * nt + i] = v;
        }
    }
    if (set_skip) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) skip[nt] = acc[nt];
    }
}

template <int DX2, int KD>
__global__ __launch_bounds__(256, (DX2 <= 128 ? 2 : 1)) void mel_decoder_kernel(const MelDecP p) {
    constexpr int NT = DX2 / 32;
    constexpr int LDSROW = DX2 + 4;
    constexpr int PAD = KD / 2;
    constexpr int KCS = DX2 / 8;
    ESMI_DYN_LDS(lds);
    float* xs = lds;                                                  // [132][LDSROW]
    float* wbuf = lds + (kDecRows + 2 * kDecPadRows) * LDSROW;        // [KD+1][DX2]: taps then bias
    int* src = reinterpret_cast<int*>(wbuf + (KD + 1) * DX2);         // [128]

    const int tid = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const int i = lane & 31, h = lane >> 5;
    const int tile = (int)blockIdx.x, b = (int)blockIdx.y;
    const int L = p.lmax_dev ? *p.lmax_dev : p.lmax_host;
    const int mlen = p.mel_len ? min(p.mel_len[b], L) : L;
    const int f_lo = tile * p.TL, f0 = f_lo - p.halo;
    const int out_hi = min(f_lo + p.TL, p.L_out);
    const int valid_end = p.apply_mask ? mlen : L;
    if (f_lo >= p.L_out) return;
    if (f_lo >= valid_end) {  // whole window is padding: the final masked_fill (or the [L, L_out) tail) zeroes it
        const int n = (out_hi - f_lo) * p.n_mel;
        float* o = p.mel + ((long)b * p.L_out + f_lo) * p.n_mel;
        for (int e = tid; e < n; e += 256) o[e] = 0.0f;
        return;
    }

    // ---- phase 0: source row of every window row, zero the LDS pad rows
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
    for (int e = tid; e < 2 * kDecPadRows * LDSROW; e += 256) {
        const int r = e / LDSROW, c = e - r * LDSROW;
        const int rr = r < kDecPadRows ? r : kDecRows + r;             // rows 0,1 and 130,131
        xs[rr * LDSROW + c] = 0.0f;
    }
    __syncthreads();

    float* xs_w = xs + (kDecPadRows + 32 * w) * LDSROW;               // this wave's 32 rows
    const int* src_w = src + 32 * w;
    const f32x4* blob4 = reinterpret_cast<const f32x4*>(p.blob);

    f32x16 acc[NT], skip[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { acc[nt] = zero16(); skip[nt] = zero16(); }

    // ---- proj: Linear(d4, dx2), K processed in chunks of DX2 channels staged through the tile
    const int nchunks = p.d4 / DX2;
    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch > 0) __syncthreads();  // previous chunk fully consumed
        for (int e = tid; e < kDecRows * (DX2 / 4); e += 256) {
            const int r = e / (DX2 / 4), q = e - r * (DX2 / 4);
            const int s = src[r];
            f32x4 v = zero4();
            if (s >= 0) v = ld4(p.x + (long)s * p.d4 + ch * DX2 + 4 * q);
            *reinterpret_cast<f32x4*>(xs + (kDecPadRows + r) * LDSROW + 4 * q) = v;
        }
        __syncthreads();
        const f32x4* bw = blob4 + (p.lay.proj_w >> 2) + (long)ch * KCS * NT * 64 + lane;
        for (int kc = 0; kc < KCS; ++kc) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(xs_w + i * LDSROW + 8 * kc + 4 * h);
            f32x4 bv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = bw[(kc * NT + nt) * 64];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32(av[s], bv[nt][s], acc[nt]);
            }
        }
    }
    __syncthreads();  // every wave finished reading the staged input
    const int n_layers = p.n_blocks * p.block_depth;
    {
        dec_epilogue<NT>(acc, skip, p.blob + p.lay.proj_b, p.blob + p.lay.proj_g, p.blob + p.lay.proj_beta, nullptr,
                         nullptr, true, xs_w, src_w, lane);
        if (n_layers > 0) {
            const float* lw = p.blob + p.lay.layer0;
            for (int e = tid; e < (KD + 1) * DX2; e += 256) wbuf[e] = lw[e];  // taps + bias are contiguous
        }
    }
    __syncthreads();

    // ---- conv layers
    for (int l = 0; l < n_layers; ++l) {
        const float* lw = p.blob + p.lay.layer0 + (long)l * p.lay.layer_stride;
        const f32x4* bw = blob4 + ((p.lay.layer0 + (long)l * p.lay.layer_stride + p.lay.l_pw) >> 2) + lane;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = zero16();
        for (int kc = 0; kc < KCS; ++kc) {
            const int c = 8 * kc + 4 * h;
            f32x4 av = *reinterpret_cast<const f32x4*>(wbuf + KD * DX2 + c);  // depthwise bias
#pragma unroll
            for (int j = 0; j < KD; ++j) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xs_w + (i + j - PAD) * LDSROW + c);
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wbuf + j * DX2 + c);
#pragma unroll
                for (int s = 0; s < 4; ++s) av[s] = fmaf(xv[s], wv[s], av[s]);
            }
            f32x4 bv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = bw[(kc * NT + nt) * 64];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32(av[s], bv[nt][s], acc[nt]);
            }
        }
        __syncthreads();  // all reads of xs / wbuf for this layer done
        const bool block_end = ((l + 1) % p.block_depth) == 0;
        const float* sk = p.blob + p.lay.skip0 + (long)(l / p.block_depth) * 2 * DX2;
        dec_epilogue<NT>(acc, skip, lw + p.lay.l_pwb, lw + p.lay.l_g, lw + p.lay.l_b, block_end ? sk : nullptr,
                         block_end ? sk + DX2 : nullptr, block_end, xs_w, src_w, lane);
        if (l + 1 < n_layers) {
            const float* nw = lw + p.lay.layer_stride;
            for (int e = tid; e < (KD + 1) * DX2; e += 256) wbuf[e] = nw[e];
        }
        __syncthreads();
    }

    // ---- mel Linear(dx2, n_mel) on skip (held in the LDS tile), masked store
    {
        f32x16 m[kMelNT];
#pragma unroll
        for (int nt = 0; nt < kMelNT; ++nt) m[nt] = zero16();
        const f32x4* bw = blob4 + (p.lay.mel_w >> 2) + lane;
        for (int kc = 0; kc < KCS; ++kc) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(xs_w + i * LDSROW + 8 * kc + 4 * h);
            f32x4 bv[kMelNT];
#pragma unroll
            for (int nt = 0; nt < kMelNT; ++nt) bv[nt] = bw[(kc * kMelNT + nt) * 64];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int nt = 0; nt < kMelNT; ++nt) m[nt] = mfma32(av[s], bv[nt][s], m[nt]);
            }
        }
        const float* mb = p.blob + p.lay.mel_b;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = f0 + 32 * w + tile_row(r, lane);
            if (f < f_lo || f >= out_hi) continue;
            float* orow = p.mel + ((long)b * p.L_out + f) * p.n_mel;
            const bool live = f < valid_end;
#pragma unroll
            for (int nt = 0; nt < kMelNT; ++nt) {
                const int col = 32 * nt + i;
                if (col < p.n_mel) orow[col] = live ? m[nt][r] + mb[col] : 0.0f;
            }
        }
    }
}

}  // namespace esmi
