// Fused Fuse + variance adaptor (layers/networks.py:189-219 and :346-384):
//     fused = mask(Linear(cat_i Up_i(Linear_i(f_i))))             Up_0 = id, Up_i = ConvTranspose1d(stride 2^i) cropped
//     3 x AcousticDecoder: y = ReLU(conv1); y = ReLU(LN1(y)); y = ReLU(conv2(y)); pred = Linear(y) [on the pre-norm2 y]
//     feat = cat[fused, Emb(bucketize(pitch)), Emb(bucketize(energy)), mask(LN2(y_dur))];  dur = clamp(mask(round(pred_dur)))
// A workgroup of nw <= 4 waves owns 32*nw consecutive positions, one 32-row MFMA tile per wave.  The predictors'
// two stacked k=3 convolutions read their neighbours' rows from two LDS tiles shared by the workgroup
// ([32*nw + 2][dim + 4]: fused features, predictor hidden); only when a sequence needs several workgroups are two
// halo rows per side recomputed.  Loads / stores are unconditional (bounds-checked buffer accesses at the ragged
// edges) and each GEMM's first weight group is requested one stage ahead -- see enc_attn_ffn.h.  Replaces 11
// launches of the unfused path.
#pragma once
#include "small_kernels.h"
#include "wave_chain.h"

namespace esmi {

struct PredW {   // conv1_w / conv2_w in MFMA B-fragment order (esmi_pack_bfrag_f32)
    const float *conv1_w, *conv1_b, *ln1_g, *ln1_b, *conv2_w, *conv2_b, *ln2_g, *ln2_b, *lin_w, *lin_b, *bins, *emb;
};

// enc_pred128.h (dim = 128: one workgroup per (utterance, predictor))
struct Pred128P {
    PredW pred[3];                  // conv1_w / conv2_w: esmi_pack_bfrag_f32 arrays (three taps each)
    const unsigned char* mask;      // (B, T) or NULL
    const float* pitch_t;           // teacher values (train = True) or NULL
    const float* energy_t;
    const int* dur_t;
    float* feat;                    // (B, T, 4 dim): channels [0, dim) are read, the rest written
    float* preds[3];                // (B, T) each
    int* pitch_idx;
    int* energy_idx;
    int* dur;
    int* cum;                       // (B, T) or NULL
    int* mel_len;                   // (B)
    int B, T;
};
// enc_ffn128.h (everything behind the attention of a C = 128, two-head, expansion-2 block: one workgroup per utterance)
struct PostAttn128P {
    const float* ctx;        // (B, N, 256) attention context (two heads x 128)
    const float* x;          // (B, N, 128) the block's input rows (residual)
    float* y1;               // (B, N, 128) scratch: LN1's output (the second residual)
    float* out;              // (B, N, 128); may be x
    const float *proj_w, *ffn_w, *mlp2_w;   // esmi_pack_bfrag_f32 arrays: (128 x 256), three taps of (256 x 128), (128 x 256)
    const float *proj_b, *ln1_g, *ln1_b, *ffn_b, *ffn_b0, *ffn_b2, *mlp2_b, *ln2_g, *ln2_b;
    const unsigned char* rowmask;            // (B, N) 1 = padding row, or NULL
    int B, N;
};
// enc_merge256.h (merge convolution stride 2, 128 -> 256, + the folded attention's query GEMM 256 -> heads x 256: one workgroup per utterance)
struct MergeQ256P {
    const float* x_in;       // (B, n_in, 128)
    float* x_out;            // (B, n_out, 256)
    float* q;                // (B, n_out, heads * 256)
    const float *merge_w, *q_w;   // esmi_pack_bfrag_f32 arrays: `kernel` taps of (256 x 128); (heads * 256 x 256)
    int B, n_in, n_out, kernel, heads;
};
struct FuseVaP {
    int B, T, depth, kernel;
    const float* feats[4];
    int n_i[4];
    const float* mlp_w[4];  // mlp_w, up_w, fuse_w: MFMA B-fragment order (esmi_pack_bfrag_f32, see wave_chain.h)
    const float* mlp_b[4];
    const float* up_w[4];   // (k, dim, dim) tap-major
    const float* up_b[4];
    const float* fuse_w;    // (dim, depth*dim)
    const float* fuse_b;
    PredW pred[3];          // pitch, energy, duration
    const unsigned char* mask;
    const float* pitch_t;
    const float* energy_t;
    const int* dur_t;
    float* feat;            // (B,T,4*dim)
    float* preds[3];        // (B,T) each
    int* pitch_idx;
    int* energy_idx;
    int* dur;
    // optional decoder head, dim == 32 only (4*dim = dx2 = 128): h0 = LN(tanh(Linear(4*dim, dx2)(feat))) at phoneme rate
    const float *head_w, *head_b, *head_g, *head_beta;   // head_w in MFMA B-fragment order
    float* h0;              // (B,T,128) or NULL
    int* cum;               // (B,T) inclusive cumsum of max(dur,0) and
    int* mel_len;           // (B) its total: written when one workgroup covers the utterance (halo == 0), else NULL
    int wgs_per_b;          // workgroups per utterance
    int useful;             // positions stored per workgroup: 32*nw - 2*halo
    int halo;               // 0: one workgroup covers the sequence, 2: two recomputed rows per side
};

constexpr int kVaMaxWaves = 4;

// LDS floats of an nw-wave workgroup: fb0 [32nw+2][dim+4], then one region shared over time by the per-wave Fuse
// scratch (cat [32][depth*dim+4] + tmp [32][dim+4] each) and the predictor-hidden tile tb0 [32nw+2][3*dim+4]
inline int fuse_va_lds_floats(int dim, int depth, int nw, bool head = false) {
    const int fbt = (32 * nw + 2) * (dim + 4), tbt = (32 * nw + 2) * (3 * dim + 4);
    const int priv = nw * (32 * (depth * dim + 4) + 32 * (dim + 4));
    const int ft = head ? nw * 32 * (4 * dim + 4) : 0;       // per-wave feature tiles of the decoder-head stage
    const int m = priv > tbt ? priv : tbt;
    return fbt + (ft > m ? ft : m);
}

inline void fuse_va_plan(int n, int dim, int depth, int* nw, int* wgs, int* useful, int* halo) {
    int nwmax = kVaMaxWaves;
    while (nwmax > 1 && fuse_va_lds_floats(dim, depth, nwmax) * 4 > 150 * 1024) --nwmax;
    if (n <= 32 * nwmax) { *nw = (n + 31) / 32; *wgs = 1; *useful = 32 * *nw; *halo = 0; return; }
    int best = 1, best_waves = 1 << 30;
    for (int w = 1; w <= nwmax; ++w) {
        const int u = 32 * w - 4, waves = ((n + u - 1) / u) * w;
        if (waves <= best_waves) { best_waves = waves; best = w; }
    }
    *nw = best; *useful = 32 * best - 4; *wgs = (n + *useful - 1) / *useful; *halo = 2;
}

#ifdef ESMI_E3_STAGED
// development only: layernorm_tile_regs with the 32-lane reductions of RB rows at a time done stage by stage (all ds_swizzle round trips
// of a batch in flight together) -- the form that produced wrong rows inside enc_fuse_va_kernel in round 4 for RB = 8 and 16
#ifndef ESMI_E3_SWZ_FENCE
#define ESMI_E3_SWZ_FENCE 0
#endif
template <int RB>
__device__ __forceinline__ void row_sum32_batch(float (&s)[RB]) {
    float t[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        s[r] += dpp_f<0xB1>(s[r]);
        s[r] += dpp_f<0x4E>(s[r]);
        s[r] += dpp_f<0x141>(s[r]);
        s[r] += dpp_f<0x140>(s[r]);
    }
#if ESMI_E3_SWZ_FENCE == 2
    asm volatile("s_nop 4" ::: "memory");
#elif ESMI_E3_SWZ_FENCE == 3
    asm volatile("" ::: "memory");          // compiler-level ordering only
#elif ESMI_E3_SWZ_FENCE == 4
    __builtin_amdgcn_sched_barrier(0);      // scheduling fence only
#elif ESMI_E3_SWZ_FENCE == 5
    asm volatile("s_nop 0");                // one wait state, no memory clobber
#endif
#pragma unroll
    for (int r = 0; r < RB; ++r) t[r] = swz_xor16_f(s[r]);
#if ESMI_E3_SWZ_FENCE >= 1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#pragma unroll
    for (int r = 0; r < RB; ++r) s[r] += t[r];
}
template <int NT, int RB>
__device__ __forceinline__ void layernorm_tile_regs_batched(f32x16 (&v)[NT], const float (&gg)[NT], const float (&bb)[NT], float eps = 1e-5f) {
    const float inv_c = 1.0f / (float)(32 * NT);
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += RB) {
        float s[RB], q[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            s[r] = 0.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) s[r] += v[nt][r0 + r];
        }
        row_sum32_batch<RB>(s);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            s[r] *= inv_c;
            q[r] = 0.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float d = v[nt][r0 + r] - s[r];
                q[r] = fmaf(d, d, q[r]);
            }
        }
        row_sum32_batch<RB>(q);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const float rstd = rsqrt_fast_f32(q[r] * inv_c + eps);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) v[nt][r0 + r] = fmaf((v[nt][r0 + r] - s[r]) * rstd, gg[nt], bb[nt]);
        }
    }
}
#endif

__device__ __forceinline__ int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

template <int ND, int KU>   // dim = 32*ND, ConvTranspose1d kernel KU
__device__ __forceinline__ void enc_fuse_va_body(const FuseVaP& p) {
    constexpr int DIM = 32 * ND, LDD = DIM + 4, LDT = 3 * DIM + 4;
    ESMI_DYN_LDS(lds);
    const int nw = (int)(blockDim.x >> 6), w = wave_id();
    const int lane = lane_id(), i = lane & 31, h2 = lane >> 5;
    const int ldc = p.depth * DIM + 4;
    float* fb0 = lds;                                 // [32nw+2][LDD] fused features, zero rows around
    float* tb0 = fb0 + (32 * nw + 2) * LDD;           // [32nw+2][LDT] hidden of the three predictors side by side, zero rows around ...
    float* cat = tb0 + w * (32 * ldc + 32 * LDD);     // ... aliased, during Fuse, by each wave's [32][ldc]
    float* tmp = cat + 32 * ldc;                      //     and [32][LDD]
    const int r0 = 32 * w;
    float* fb = fb0 + LDD * (1 + r0);
    float* tb = tb0 + LDT * (1 + r0);
    const int b = (int)blockIdx.x / p.wgs_per_b, wg = (int)blockIdx.x - b * p.wgs_per_b;
    const int p0 = wg * p.useful - p.halo + r0;       // position of this wave's row 0
    for (int c = (int)threadIdx.x; c < LDD; c += (int)blockDim.x) {
        fb0[c] = 0.0f;
        fb0[(32 * nw + 1) * LDD + c] = 0.0f;
    }
    const int pos_i = p0 + i;
    const bool in_i = pos_i >= 0 && pos_i < p.T;
    const BufRsrc r_mask = make_rsrc(p.mask ? p.mask + (long)b * p.T : nullptr, p.T);
    const BufRsrc r_feat = make_rsrc(p.feat + (long)b * p.T * 4 * DIM, (long)p.T * 4 * DIM * 4);
    const unsigned mb = buf_ld_u8(r_mask, (unsigned)pos_i);   // rows outside [0, T) read 0

    ESMI_CT_INIT(2);
    ESMI_CT();   // 0
    // ---------------- Fuse
    f32x16 a[ND];
    WaveGrp<ND> gw;
    {   // level 0: Linear(dim, dim) on f_0 rows
        wave_prefetch<ND>(gw, p.mlp_w[0], ND, 0, 0, lane);
        zero_tiles<ND>(a);
        const float* arow = p.feats[0] + ((long)b * p.n_i[0] + (in_i ? pos_i : 0)) * DIM + 4 * h2;
        wave_gemm_k<ND, ND>(a, gw, arow, in_i, p.mlp_w[0], ND, 0, 0, lane);
        if (p.depth > 1) wave_prefetch<ND>(gw, p.mlp_w[1], ND, 0, 0, lane);
        else wave_prefetch<ND>(gw, p.fuse_w, ND, 0, 0, lane);
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
            const float bc = p.mlp_b[0][32 * nt + i];
#pragma unroll
            for (int r = 0; r < 16; ++r) a[nt][r] += bc;
        }
        tile_store<ND>(cat, ldc, 0, a, lane);
        ESMI_CT();   // level 0 done
    }
    for (int lv = 1; lv < p.depth; ++lv) {   // Linear(dim*2^lv, dim) -> ConvTranspose1d(stride 2^lv), cropped to T
        const int s = 1 << lv, cl = DIM << lv, nl = p.n_i[lv];
        const int n_base = floor_div(p0 - (KU - 1), s);
        zero_tiles<ND>(a);
        const int n = n_base + i;
        const bool n_ok = n >= 0 && n < nl;
        const float* arow = p.feats[lv] + ((long)b * nl + (n_ok ? n : 0)) * cl + 4 * h2;
        wave_gemm<ND>(a, gw, arow, n_ok, cl, p.mlp_w[lv], ND, 0, 0, lane);
        wave_prefetch<ND>(gw, p.up_w[lv], ND, 0, 0, lane);
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
            const float bc = p.mlp_b[lv][32 * nt + i];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nr = n_base + tile_row(r, lane);
                a[nt][r] = (nr >= 0 && nr < nl) ? a[nt][r] + bc : 0.0f;   // rows that do not exist contribute nothing
            }
        }
        lds_wave_sync();
        tile_store<ND>(tmp, LDD, 0, a, lane);
        lds_wave_sync();
        ESMI_CT();   // level linear done
        zero_tiles<ND>(a);
        {   // out[n*s + j] += in[n] W[:, :, j]
            const float* taps[KU];
            bool tok[KU];
#pragma unroll
            for (int j = 0; j < KU; ++j) {
                const int q = pos_i - j;
                const int nq = q / s;
                tok[j] = q >= 0 && (q - nq * s) == 0 && nq < nl;
                taps[j] = tmp + (tok[j] ? nq - n_base : 0) * LDD + 4 * h2;
            }
            wave_gemm_taps<ND, KU, ND, true>(a, gw, taps, tok, p.up_w[lv], (long)DIM * DIM, ND, 0, 0, lane);
        }
        if (lv + 1 < p.depth) wave_prefetch<ND>(gw, p.mlp_w[lv + 1], ND, 0, 0, lane);
        else wave_prefetch<ND>(gw, p.fuse_w, ND, 0, 0, lane);
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
            const float bc = p.up_b[lv][32 * nt + i];
#pragma unroll
            for (int r = 0; r < 16; ++r) a[nt][r] += bc;
        }
        tile_store<ND>(cat, ldc, lv * DIM, a, lane);
        ESMI_CT();   // up-conv done
    }
    lds_wave_sync();
    zero_tiles<ND>(a);
    wave_gemm<ND>(a, gw, cat + i * ldc + 4 * h2, true, p.depth * DIM, p.fuse_w, ND, 0, 0, lane);

    bool rout[16], rz[16], live[16];        // per accumulator row: outside the sequence / masked (padding) / stored by this workgroup
    int rpos[16];
    {
        const unsigned mbits = (unsigned)ballot64(mb != 0);   // bit i = row i (both half waves hold the same rows)
        const int row_lo = p.halo, row_hi = 32 * nw - p.halo;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tile_row(r, lane);
            rpos[r] = p0 + row;
            rout[r] = rpos[r] < 0 || rpos[r] >= p.T;
            rz[r] = !rout[r] && ((mbits >> row) & 1u);
            live[r] = r0 + row >= row_lo && r0 + row < row_hi && !rout[r];
        }
    }
#pragma unroll
    for (int nt = 0; nt < ND; ++nt) {
        const int col = 32 * nt + i;
        const float bc = p.fuse_b[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = (rout[r] || rz[r]) ? 0.0f : a[nt][r] + bc;   // masked_fill; outside rows = conv zero padding
            a[nt][r] = v;
            buf_st(r_feat, live[r] ? (unsigned)((rpos[r] * 4 * DIM + col) * 4) : kBufOOB, v);
        }
    }
    tile_store<ND>(fb, LDD, 0, a, lane);
    __syncthreads();            // Fuse scratch is dead in every wave; fused rows of the neighbours are in place
    for (int c = (int)threadIdx.x; c < LDT; c += (int)blockDim.x) {   // tb0's zero rows (they alias wave 0's / the last wave's scratch)
        tb0[c] = 0.0f;
        tb0[(32 * nw + 1) * LDT + c] = 0.0f;
    }

    ESMI_CT();   // 1 fuse done
    // ---------------- three predictors, side by side: one barrier pair instead of three, three independent MFMA /
    // LayerNorm chains to interleave, the embedding gathers of pitch / energy overlap the duration tail
    const float* f_row = fb + i * LDD + 4 * h2;
    const float* t_row = tb + i * LDT + 4 * h2;
    const BufRsrc r_pt = make_rsrc(p.pitch_t ? p.pitch_t + (long)b * p.T : nullptr, (long)p.T * 4);
    const BufRsrc r_et = make_rsrc(p.energy_t ? p.energy_t + (long)b * p.T : nullptr, (long)p.T * 4);
    const BufRsrc r_dt = make_rsrc(p.dur_t ? p.dur_t + (long)b * p.T : nullptr, (long)p.T * 4);
    const BufRsrc r_dur = make_rsrc(p.dur + (long)b * p.T, (long)p.T * 4);
    int* sdur = reinterpret_cast<int*>(fb0);   // durations of the workgroup's rows (fb0 is dead after conv1)
    // every small parameter is requested up front: one memory round trip
    float b1[3][ND], g1[3][ND], be1[3][ND], b2[3][ND], lw[3][ND], g2[ND], be2[ND], lb[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const PredW& w_ = p.pred[q];
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
            const int col = 32 * nt + i;
            b1[q][nt] = w_.conv1_b[col]; g1[q][nt] = w_.ln1_g[col]; be1[q][nt] = w_.ln1_b[col];
            b2[q][nt] = w_.conv2_b[col]; lw[q][nt] = w_.lin_w[col];
        }
        lb[q] = w_.lin_b[0];
    }
#pragma unroll
    for (int nt = 0; nt < ND; ++nt) {
        g2[nt] = p.pred[2].ln2_g[32 * nt + i];
        be2[nt] = p.pred[2].ln2_b[32 * nt + i];
    }
    // bucket edges replicated in both half waves: lane l holds edges (l&31) + 32*e; +inf beyond the dim-1 edges
    float edge[2][ND];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int e = 0; e < ND; ++e) {
            const int ei = 32 * e + i;
            const float ev = p.pred[q].bins[ei < DIM - 1 ? ei : DIM - 2];   // clamped: no branch around the load
            edge[q][e] = ei < DIM - 1 ? ev : INFINITY;
        }
    }
    float tv[3][16];            // teacher values of this lane's rows (0 when absent; duration: int bits)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned off = rout[r] ? kBufOOB : (unsigned)(rpos[r] * 4);
        tv[0][r] = buf_ld(r_pt, off);
        tv[1][r] = buf_ld(r_et, off);
        tv[2][r] = buf_ld(r_dt, off);
    }

    // one k=3 conv step for the three predictors: A rows either shared (conv1: the fused features) or one column
    // block per predictor (conv2: its own hidden); weights from the three packed matrices
    struct TriGrp { f32x4 a[3][4]; f32x4 b[3][4][ND]; };
    auto tri_fetch = [&](TriGrp& gq, const float* arow, int a_qstride, const float* const (&wq)[3], int tap, int g)
        __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float* wl = wq[q] + (long)tap * DIM * DIM + (long)(4 * g) * ND * 256 + 4 * lane;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                if (q == 0 || a_qstride) gq.a[q][kq] = ld4(arow + q * a_qstride + 32 * g + 8 * kq);
#pragma unroll
                for (int nt = 0; nt < ND; ++nt) gq.b[q][kq][nt] = ld4(wl + (kq * ND + nt) * 256);
            }
        }
    };
    auto tri_mma = [&](f32x16 (&acc)[3][ND], const TriGrp& gq, bool shared_a) __attribute__((always_inline)) {
#if ESMI_CHAIN_SPLIT
#pragma unroll
        for (int st = 0; st < 2; ++st) {      // split-f16x2: see wave_grp_mma
            f16x2p a2 = split_f16x2(gq.a[0][2 * st], gq.a[0][2 * st + 1]);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (q > 0 && !shared_a) a2 = split_f16x2(gq.a[q][2 * st], gq.a[q][2 * st + 1]);
#pragma unroll
                for (int nt = 0; nt < ND; ++nt)
                    acc[q][nt] = mfma32_split2(a2, __builtin_bit_cast(u32x4, gq.b[q][2 * st][nt]),
                                               __builtin_bit_cast(u32x4, gq.b[q][2 * st + 1][nt]), acc[q][nt]);
            }
        }
#else
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
#pragma unroll
                    for (int nt = 0; nt < ND; ++nt)
                        acc[q][nt] = mfma32(gq.a[shared_a ? 0 : q][kq][s4], gq.b[q][kq][nt][s4], acc[q][nt]);
                }
            }
        }
#endif
    };
    auto tri_conv = [&](f32x16 (&acc)[3][ND], const float* row, int ldrow, int a_qstride, const float* const (&wq)[3])
        __attribute__((always_inline)) {
        constexpr int STEPS = 3 * ND;   // tap-major: step n = tap*ND + g
        TriGrp ga, gb;
        tri_fetch(ga, row - ldrow, a_qstride, wq, 0, 0);
#pragma unroll
        for (int n = 0; n < STEPS; ++n) {
            const int tn = (n + 1) / ND, gn = (n + 1) % ND;
            if (n + 1 < STEPS) {
                if (n & 1) tri_fetch(ga, row + (tn - 1) * ldrow, a_qstride, wq, tn, gn);
                else tri_fetch(gb, row + (tn - 1) * ldrow, a_qstride, wq, tn, gn);
            }
            sched_fence();
            if (n & 1) tri_mma(acc, gb, a_qstride == 0);
            else tri_mma(acc, ga, a_qstride == 0);
            sched_fence();
        }
#if ESMI_CHAIN_SPLIT
#pragma unroll
        for (int q = 0; q < 3; ++q) {   // acc was zero at the start (both call sites): take the weights' 2^8 out
#pragma unroll
            for (int nt = 0; nt < ND; ++nt) acc[q][nt] *= kF16WScaleInv;
        }
#endif
    };

    f32x16 c[3][ND];
#pragma unroll
    for (int q = 0; q < 3; ++q) zero_tiles<ND>(c[q]);
    {
        const float* const w1[3] = {p.pred[0].conv1_w, p.pred[1].conv1_w, p.pred[2].conv1_w};
        tri_conv(c, f_row, LDD, 0, w1);
    }
    ESMI_CT();   // conv1 issued
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c[q][nt][r] = fmaxf(c[q][nt][r] + b1[q][nt], 0.0f);
        }
#ifdef ESMI_E3_STAGED   // development only (profiles/r05_probes/fuse_va_wrong_rows.md): the round-4 experiment that gave wrong rows on the GPU
        layernorm_tile_regs_batched<ND, ESMI_E3_STAGED>(c[q], g1[q], be1[q]);
#else
        layernorm_tile_regs<ND>(c[q], g1[q], be1[q]);
#endif
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c[q][nt][r] = rout[r] ? 0.0f : fmaxf(c[q][nt][r], 0.0f);
        }
        tile_store<ND>(tb, LDT, q * DIM, c[q], lane);
        zero_tiles<ND>(c[q]);
    }
    ESMI_CT();   // LN1 + store
    __syncthreads();   // neighbours' hidden rows (and tb0's zero rows) are in place
    {
        const float* const w2[3] = {p.pred[0].conv2_w, p.pred[1].conv2_w, p.pred[2].conv2_w};
        tri_conv(c, t_row, LDT, DIM, w2);
    }
    ESMI_CT();   // conv2 issued
    float pr[3][16];   // Linear(dim, 1) on the pre-norm2 tensor (networks.py:157-160)
    int bidx[2][16];   // torch.bucketize(v, edges, right=False) = number of edges strictly below v
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c[q][nt][r] = fmaxf(c[q][nt][r] + b2[q][nt], 0.0f);
        }
        const bool has_t = q == 0 ? p.pitch_t != nullptr : p.energy_t != nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = 0.0f;
#pragma unroll
            for (int nt = 0; nt < ND; ++nt) s = fmaf(c[q][nt][r], lw[q][nt], s);
            s = row_sum32(s) + lb[q];
            if (q == 2) s = fmaxf(s, 0.0f);
            pr[q][r] = s;
            if (q < 2) {   // compile-time branch; the ballots are executed by all lanes
                const float v = (has_t && !rout[r]) ? tv[q][r] : s;
                int bi = 0;
#pragma unroll
                for (int e = 0; e < ND; ++e) {
                    const unsigned long long m = ballot64(edge[q][e] < v);
                    bi += __builtin_popcount((unsigned)(h2 ? (m >> 32) : m));
                }
                bidx[q][r] = bi;
            }
        }
    }
    ESMI_CT();   // dots done
    float emb[2][16][ND];      // pitch / energy embedding rows: gathers in flight during the duration tail
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int nt = 0; nt < ND; ++nt) emb[q][r][nt] = p.pred[q].emb[bidx[q][r] * DIM + 32 * nt + i];
        }
    }
    layernorm_tile_regs<ND>(c[2], g2, be2);   // duration features (networks.py:161-163)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const BufRsrc r_pred = make_rsrc(p.preds[q] + (long)b * p.T, (long)p.T * 4);
        int* const ip = q == 0 ? p.pitch_idx : p.energy_idx;
        const BufRsrc r_idx = make_rsrc((q < 2 && ip) ? ip + (long)b * p.T : nullptr, (long)p.T * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned frow = live[r] ? (unsigned)((rpos[r] * 4 * DIM + (q == 2 ? 3 : 1 + q) * DIM + i) * 4) : kBufOOB;
            const unsigned srow = (live[r] && i == 0) ? (unsigned)(rpos[r] * 4) : kBufOOB;   // one lane per row
            buf_st(r_pred, srow, pr[q][r]);
            if (q == 2) {
#pragma unroll
                for (int nt = 0; nt < ND; ++nt) buf_st(r_feat, frow + 128u * nt, rz[r] ? 0.0f : c[2][nt][r]);
                float d = p.dur_t ? (float)__builtin_bit_cast(int, tv[2][r]) : rintf(pr[2][r]);   // torch.round: half to even
                if (p.mask) {                                                                        // networks.py:381-382
                    if (rz[r]) d = 0.0f;
                    d = fmaxf(d, 0.0f);
                }
                buf_st_i(r_dur, srow, (int)d);
                if (p.cum && i == 0) sdur[r0 + tile_row(r, lane)] = rout[r] ? 0 : max((int)d, 0);
            } else {
#pragma unroll
                for (int nt = 0; nt < ND; ++nt) buf_st(r_feat, frow + 128u * nt, rz[r] ? 0.0f : emb[q][r][nt]);
                buf_st_i(r_idx, srow, bidx[q][r]);
            }
        }
    }
    ESMI_CT();   // outputs done
    if (ND == 1 && p.h0) {
        // ---------------- decoder head at phoneme rate: h0 = LN(tanh(feat . Wp^T + b)).  MelDecoder's first stage is
        // row-wise and every frame of a phoneme reads the same row, so it is computed here once per phoneme while the
        // 4*dim features are still in registers; the decoder then only gathers (mel_decoder.h).
        constexpr int D4 = 4 * DIM, LDF = D4 + 4;
        float* ft = tb0 + w * (32 * LDF);       // [32][LDF], aliases the predictor-hidden tile
        WaveGrp<4> gh;
        wave_prefetch<4>(gh, p.head_w, 4, 0, 0, lane);
        float hb[4], hg[4], hbe[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            hb[nt] = p.head_b[32 * nt + i]; hg[nt] = p.head_g[32 * nt + i]; hbe[nt] = p.head_beta[32 * nt + i];
        }
        __syncthreads();                        // every wave is done with tb0 (conv2) and fb0
        f32x16 fe[1];
#pragma unroll
        for (int part = 0; part < 4; ++part) {  // the row of feat exactly as stored: fused | pitch emb | energy emb | duration feats
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = part == 0 ? a[0][r] : (part == 1 ? emb[0][r][0] : (part == 2 ? emb[1][r][0] : c[2][0][r]));
                fe[0][r] = (part > 0 && rz[r]) ? 0.0f : v;
            }
            tile_store<1>(ft, LDF, part * DIM, fe, lane);
        }
        lds_wave_sync();
        f32x16 hh[4];
        zero_tiles<4>(hh);
        wave_gemm_k<4, 4>(hh, gh, ft + i * LDF + 4 * h2, true, p.head_w, 4, 0, 0, lane);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) hh[nt][r] = tanh_fast_f32(hh[nt][r] + hb[nt]);
        }
        layernorm_tile_regs<4>(hh, hg, hbe);
        const BufRsrc r_h0 = make_rsrc(p.h0 + (long)b * p.T * D4, (long)p.T * D4 * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned off = live[r] ? (unsigned)((rpos[r] * D4 + i) * 4) : kBufOOB;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) buf_st(r_h0, off + 128u * nt, hh[nt][r]);
        }
    }
    if (p.cum) {   // FeatureUpsampler's scan (networks.py:233-244) while the durations are still on the CU; T <= 128 here
        __syncthreads();
        if (w == 0) {
            const int per = (p.T + 63) / 64, q0 = lane * per;
            int local = 0;
            for (int q = 0; q < per; ++q) local += (q0 + q < p.T) ? sdur[q0 + q] : 0;
            int incl = local;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = shfl_up_i(incl, d);
                if (lane >= d) incl += v;
            }
            const BufRsrc r_cum = make_rsrc(p.cum + (long)b * p.T, (long)p.T * 4);
            int run = incl - local;
            for (int q = 0; q < per; ++q) {
                run += (q0 + q < p.T) ? sdur[q0 + q] : 0;
                buf_st_i(r_cum, (unsigned)((q0 + q) * 4), run);      // positions >= T fall off the buffer end
            }
            const int total = shfl_i(incl, 63);
            if (lane == 0) p.mel_len[b] = total;
        }
    }
}

template <int ND, int KU>
__global__ __launch_bounds__(64 * kVaMaxWaves, kChainWps) void enc_fuse_va_kernel(const FuseVaP p) {
    enc_fuse_va_body<ND, KU>(p);
}


}  // namespace esmi
