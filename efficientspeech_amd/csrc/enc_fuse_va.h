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

#ifndef ESMI_E3_WPS
#define ESMI_E3_WPS ESMI_CHAIN_WPS
#endif

namespace esmi {

struct PredW {   // conv1_w / conv2_w in MFMA B-fragment order (esmi_pack_bfrag_f32)
    const float *conv1_w, *conv1_b, *ln1_g, *ln1_b, *conv2_w, *conv2_b, *ln2_g, *ln2_b, *lin_w, *lin_b, *bins, *emb;
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
    int wgs_per_b;          // workgroups per utterance
    int useful;             // positions stored per workgroup: 32*nw - 2*halo
    int halo;               // 0: one workgroup covers the sequence, 2: two recomputed rows per side
};

constexpr int kVaMaxWaves = 4;

// LDS floats of an nw-wave workgroup: fb0 [32nw+2][dim+4], then one region shared over time by the per-wave Fuse
// scratch (cat [32][depth*dim+4] + tmp [32][dim+4] each) and the predictor-hidden tile tb0 [32nw+2][dim+4]
inline int fuse_va_lds_floats(int dim, int depth, int nw) {
    const int shared = (32 * nw + 2) * (dim + 4), priv = nw * (32 * (depth * dim + 4) + 32 * (dim + 4));
    return shared + (priv > shared ? priv : shared);
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

__device__ __forceinline__ int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

template <int ND>   // dim = 32*ND
__global__ __launch_bounds__(64 * kVaMaxWaves, ESMI_E3_WPS) void enc_fuse_va_kernel(const FuseVaP p) {
    constexpr int DIM = 32 * ND, LDD = DIM + 4;
    ESMI_DYN_LDS(lds);
    const int nw = (int)(blockDim.x >> 6), w = wave_id();
    const int lane = lane_id(), i = lane & 31, h2 = lane >> 5;
    const int ldc = p.depth * DIM + 4;
    float* fb0 = lds;                                 // [32nw+2][LDD] fused features, zero rows around
    float* tb0 = fb0 + (32 * nw + 2) * LDD;           // [32nw+2][LDD] predictor hidden, zero rows around ...
    float* cat = tb0 + w * (32 * ldc + 32 * LDD);     // ... aliased, during Fuse, by each wave's [32][ldc]
    float* tmp = cat + 32 * ldc;                      //     and [32][LDD]
    const int r0 = 32 * w;
    float* fb = fb0 + LDD * (1 + r0);
    float* tb = tb0 + LDD * (1 + r0);
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
    }
    for (int lv = 1; lv < p.depth; ++lv) {   // Linear(dim*2^lv, dim) -> ConvTranspose1d(stride 2^lv), cropped to T
        const int s = 1 << lv, cl = DIM << lv, nl = p.n_i[lv];
        const int n_base = floor_div(p0 - (p.kernel - 1), s);
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
        zero_tiles<ND>(a);
        {   // out[n*s + j] += in[n] W[:, :, j]
            const float* taps[7];
            bool tok[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int q = pos_i - j;
                const int nq = q / s;
                tok[j] = j < p.kernel && q >= 0 && (q - nq * s) == 0 && nq < nl;
                taps[j] = tmp + (tok[j] ? nq - n_base : 0) * LDD + 4 * h2;
            }
            wave_gemm_taps<ND, 7, ND, true>(a, gw, taps, tok, p.kernel, p.up_w[lv], (long)DIM * DIM, ND, 0, 0, lane);
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
    }
    lds_wave_sync();
    zero_tiles<ND>(a);
    wave_gemm<ND>(a, gw, cat + i * ldc + 4 * h2, true, p.depth * DIM, p.fuse_w, ND, 0, 0, lane);
    wave_prefetch<ND>(gw, p.pred[0].conv1_w, ND, 0, 0, lane);

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
    for (int c = (int)threadIdx.x; c < LDD; c += (int)blockDim.x) {   // tb0's zero rows (they alias wave 0's / the last wave's scratch)
        tb0[c] = 0.0f;
        tb0[(32 * nw + 1) * LDD + c] = 0.0f;
    }

    ESMI_CT();   // 1 fuse done
    // ---------------- three predictors
    const float* f_row = fb + i * LDD + 4 * h2;
    const float* t_row = tb + i * LDD + 4 * h2;
    const bool tok3[3] = {true, true, true};   // the tiles carry their own zero rows
    const BufRsrc r_pt = make_rsrc(p.pitch_t ? p.pitch_t + (long)b * p.T : nullptr, (long)p.T * 4);
    const BufRsrc r_et = make_rsrc(p.energy_t ? p.energy_t + (long)b * p.T : nullptr, (long)p.T * 4);
    const BufRsrc r_dt = make_rsrc(p.dur_t ? p.dur_t + (long)b * p.T : nullptr, (long)p.T * 4);
    const BufRsrc r_dur = make_rsrc(p.dur + (long)b * p.T, (long)p.T * 4);
#pragma unroll
    for (int q = 0; q < 3; ++q) {   // unrolled: q is a compile-time constant below
        const PredW& w_ = p.pred[q];
        // every small parameter of this predictor is requested up front: one memory round trip instead of six
        float b1[ND], g1[ND], be1[ND], b2[ND], lw[ND], g2[ND], be2[ND];
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
            const int col = 32 * nt + i;
            b1[nt] = w_.conv1_b[col]; g1[nt] = w_.ln1_g[col]; be1[nt] = w_.ln1_b[col];
            b2[nt] = w_.conv2_b[col]; lw[nt] = w_.lin_w[col];
            g2[nt] = q == 2 ? w_.ln2_g[col] : 0.0f; be2[nt] = q == 2 ? w_.ln2_b[col] : 0.0f;
        }
        const float lb = w_.lin_b[0];
        // bucket edges replicated in both half waves: lane l holds edges (l&31) + 32*e; +inf beyond the dim-1 edges
        float edge[ND];
#pragma unroll
        for (int e = 0; e < ND; ++e) {
            const int ei = 32 * e + i;
            edge[e] = INFINITY;
            if (q < 2) {
                const float ev = w_.bins[ei < DIM - 1 ? ei : DIM - 2];   // clamped: no branch around the load
                if (ei < DIM - 1) edge[e] = ev;
            }
        }
        float tv[16];           // teacher values of this lane's rows (0 when absent)
        {
            const BufRsrc& rt = q == 0 ? r_pt : (q == 1 ? r_et : r_dt);
#pragma unroll
            for (int r = 0; r < 16; ++r) tv[r] = buf_ld(rt, rout[r] ? kBufOOB : (unsigned)(rpos[r] * 4));
        }
        f32x16 c[ND];
        zero_tiles<ND>(c);
        {
            const float* const taps[3] = {f_row - LDD, f_row, f_row + LDD};
            wave_gemm_taps<ND, 3, ND, false>(c, gw, taps, tok3, 3, w_.conv1_w, (long)DIM * DIM, ND, 0, 0, lane);
        }
        wave_prefetch<ND>(gw, w_.conv2_w, ND, 0, 0, lane);
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c[nt][r] = fmaxf(c[nt][r] + b1[nt], 0.0f);
        }
        ESMI_CT();   // conv1 done
        layernorm_tile_regs<ND>(c, g1, be1);
        ESMI_CT();   // LN1 done
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c[nt][r] = rout[r] ? 0.0f : fmaxf(c[nt][r], 0.0f);
        }
        __syncthreads();   // every wave finished reading tb (previous predictor's conv2)
        tile_store<ND>(tb, LDD, 0, c, lane);
        __syncthreads();   // neighbours' hidden rows are in place
        zero_tiles<ND>(c);
        {
            const float* const taps[3] = {t_row - LDD, t_row, t_row + LDD};
            wave_gemm_taps<ND, 3, ND, false>(c, gw, taps, tok3, 3, w_.conv2_w, (long)DIM * DIM, ND, 0, 0, lane);
        }
        if (q < 2) wave_prefetch<ND>(gw, p.pred[q + 1].conv1_w, ND, 0, 0, lane);
        ESMI_CT();   // conv2 done
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c[nt][r] = fmaxf(c[nt][r] + b2[nt], 0.0f);
        }
        float pr[16];      // Linear(dim, 1) on the pre-norm2 tensor (networks.py:157-160)
        int bidx[16];      // torch.bucketize(v, edges, right=False) = number of edges strictly below v
        const bool has_t = q == 0 ? p.pitch_t != nullptr : p.energy_t != nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = 0.0f;
#pragma unroll
            for (int nt = 0; nt < ND; ++nt) s = fmaf(c[nt][r], lw[nt], s);
            s = row_sum32(s) + lb;
            if (q == 2) s = fmaxf(s, 0.0f);
            pr[r] = s;
            bidx[r] = 0;
            if (q < 2) {   // wave-uniform branch: the ballots below are executed by all lanes
                const float v = (has_t && !rout[r]) ? tv[r] : s;
#pragma unroll
                for (int e = 0; e < ND; ++e) {
                    const unsigned long long m = ballot64(edge[e] < v);
                    bidx[r] += __builtin_popcount((unsigned)(h2 ? (m >> 32) : m));
                }
            }
        }
        ESMI_CT();   // dot done
        if (q == 2) layernorm_tile_regs<ND>(c, g2, be2);   // duration features (networks.py:161-163)
        const BufRsrc r_pred = make_rsrc(p.preds[q] + (long)b * p.T, (long)p.T * 4);
        int* const ip = q == 0 ? p.pitch_idx : p.energy_idx;
        const BufRsrc r_idx = make_rsrc((q < 2 && ip) ? ip + (long)b * p.T : nullptr, (long)p.T * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned frow = live[r] ? (unsigned)((rpos[r] * 4 * DIM + (q == 2 ? 3 : 1 + q) * DIM + i) * 4) : kBufOOB;
            const unsigned srow = (live[r] && i == 0) ? (unsigned)(rpos[r] * 4) : kBufOOB;   // one lane per row
            buf_st(r_pred, srow, pr[r]);
            if (q == 2) {
#pragma unroll
                for (int nt = 0; nt < ND; ++nt) buf_st(r_feat, frow + 128u * nt, rz[r] ? 0.0f : c[nt][r]);
                float d = p.dur_t ? (float)__builtin_bit_cast(int, tv[r]) : rintf(pr[r]);   // torch.round: half to even
                if (p.mask) {                                                                  // networks.py:381-382
                    if (rz[r]) d = 0.0f;
                    d = fmaxf(d, 0.0f);
                }
                buf_st_i(r_dur, srow, (int)d);
            } else {
#pragma unroll
                for (int nt = 0; nt < ND; ++nt)
                    buf_st(r_feat, frow + 128u * nt, rz[r] ? 0.0f : w_.emb[bidx[r] * DIM + 32 * nt + i]);
                buf_st_i(r_idx, srow, bidx[r]);
            }
        }
        ESMI_CT();   // outputs done
    }
}

}  // namespace esmi
