// Fused Fuse + variance adaptor (layers/networks.py:189-219 and :346-384):
//     fused = mask(Linear(cat_i Up_i(Linear_i(f_i))))             Up_0 = id, Up_i = ConvTranspose1d(stride 2^i) cropped
//     3 x AcousticDecoder: y = ReLU(conv1); y = ReLU(LN1(y)); y = ReLU(conv2(y)); pred = Linear(y) [on the pre-norm2 y]
//     feat = cat[fused, Emb(bucketize(pitch)), Emb(bucketize(energy)), mask(LN2(y_dur))];  dur = clamp(mask(round(pred_dur)))
// one wave per (utterance, 28-position tile): two halo rows per side cover the predictors' two stacked k=3
// convolutions.  Replaces 11 launches of the unfused path.
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
    int tiles_per_b;        // ceil(T / 28)
};

constexpr int kVaTileRows = 28;

__host__ __device__ inline int fuse_va_lds_floats(int dim, int depth) {
    return 32 * (depth * dim + 4) + 32 * (dim + 4) + 2 * 34 * (dim + 4);
}

__device__ __forceinline__ int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

template <int ND>   // dim = 32*ND
__global__ __launch_bounds__(64, ESMI_E3_WPS) void enc_fuse_va_kernel(const FuseVaP p) {
    constexpr int DIM = 32 * ND, LDD = DIM + 4;
    ESMI_DYN_LDS(lds);
    const int ldc = p.depth * DIM + 4;
    float* cat = lds;                       // [32][ldc]
    float* tmp = cat + 32 * ldc;            // [32][LDD]
    float* fb0 = tmp + 32 * LDD;            // [34][LDD] fused features, zero rows around
    float* tb0 = fb0 + 34 * LDD;            // [34][LDD] predictor hidden, zero rows around
    float* fb = fb0 + LDD;
    float* tb = tb0 + LDD;
    const int lane = lane_id(), i = lane & 31, h2 = lane >> 5;
    const int b = (int)blockIdx.x / p.tiles_per_b, tile = (int)blockIdx.x - b * p.tiles_per_b;
    const int p0 = tile * kVaTileRows - 2;  // position of tile row 0
    for (int c = lane; c < LDD; c += 64) {
        fb0[c] = 0.0f; fb0[33 * LDD + c] = 0.0f;
        tb0[c] = 0.0f; tb0[33 * LDD + c] = 0.0f;
    }
    const int pos_i = p0 + i;
    const bool in_i = pos_i >= 0 && pos_i < p.T;
    bool rout[16], rz[16];                  // per accumulator row: outside the sequence / masked (padding)
    int rpos[16];
    {   // ONE mask byte per lane (row i) + a ballot, instead of 16 dependent byte loads per lane
        const unsigned char mb = (in_i && p.mask) ? p.mask[(long)b * p.T + pos_i] : (unsigned char)0;
        const unsigned mbits = (unsigned)ballot64(mb != 0);   // bit i = row i (both half waves hold the same rows)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tile_row(r, lane);
            rpos[r] = p0 + row;
            rout[r] = rpos[r] < 0 || rpos[r] >= p.T;
            rz[r] = !rout[r] && ((mbits >> row) & 1u);
        }
    }

    ESMI_CT_INIT(2);
    ESMI_CT();   // 0
    // ---------------- Fuse
    f32x16 a[ND];
    {   // level 0: Linear(dim, dim) on f_0 rows
        zero_tiles<ND>(a);
        const float* arow = p.feats[0] + ((long)b * p.n_i[0] + (in_i ? pos_i : 0)) * DIM + 4 * h2;
        WaveGrp<ND> gw;
        wave_prefetch<ND>(gw, p.mlp_w[0], ND, 0, 0, lane);
        wave_gemm<ND>(a, gw, arow, in_i, DIM, p.mlp_w[0], ND, 0, 0, lane);
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
        WaveGrp<ND> gw;
        wave_prefetch<ND>(gw, p.mlp_w[lv], ND, 0, 0, lane);
        wave_gemm<ND>(a, gw, arow, n_ok, cl, p.mlp_w[lv], ND, 0, 0, lane);
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
            const float bc = p.mlp_b[lv][32 * nt + i];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nr = n_base + tile_row(r, lane);
                a[nt][r] = (nr >= 0 && nr < nl) ? a[nt][r] + bc : 0.0f;   // rows that do not exist contribute nothing
            }
        }
        __syncthreads();
        tile_store<ND>(tmp, LDD, 0, a, lane);
        __syncthreads();
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
            WaveGrp<ND> gu;
            wave_prefetch<ND>(gu, p.up_w[lv], ND, 0, 0, lane);
            wave_gemm_taps<ND, 7, ND, true>(a, gu, taps, tok, p.kernel, p.up_w[lv], (long)DIM * DIM, ND, 0, 0, lane);
        }
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
            const float bc = p.up_b[lv][32 * nt + i];
#pragma unroll
            for (int r = 0; r < 16; ++r) a[nt][r] += bc;
        }
        tile_store<ND>(cat, ldc, lv * DIM, a, lane);
    }
    __syncthreads();
    zero_tiles<ND>(a);
    {
        WaveGrp<ND> gw;
        wave_prefetch<ND>(gw, p.fuse_w, ND, 0, 0, lane);
        wave_gemm<ND>(a, gw, cat + i * ldc + 4 * h2, true, p.depth * DIM, p.fuse_w, ND, 0, 0, lane);
    }
#pragma unroll
    for (int nt = 0; nt < ND; ++nt) {
        const int col = 32 * nt + i;
        const float bc = p.fuse_b[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = (rout[r] || rz[r]) ? 0.0f : a[nt][r] + bc;   // masked_fill; outside rows = conv zero padding
            a[nt][r] = v;
            const int row = tile_row(r, lane);
            if (row >= 2 && row < 2 + kVaTileRows && !rout[r]) p.feat[((long)b * p.T + rpos[r]) * 4 * DIM + col] = v;
        }
    }
    tile_store<ND>(fb, LDD, 0, a, lane);
    __syncthreads();

    ESMI_CT();   // 1 fuse done
    // ---------------- three predictors
    const float* f_row = fb + i * LDD + 4 * h2;
    const float* t_row = tb + i * LDD + 4 * h2;
    const bool tok3[3] = {true, true, true};   // the tiles carry their own zero rows
    for (int q = 0; q < 3; ++q) {
        const PredW& w = p.pred[q];
        // every small parameter of this predictor is requested up front: one memory round trip instead of six
        float b1[ND], g1[ND], be1[ND], b2[ND], lw[ND], g2[ND], be2[ND];
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
            const int col = 32 * nt + i;
            b1[nt] = w.conv1_b[col]; g1[nt] = w.ln1_g[col]; be1[nt] = w.ln1_b[col];
            b2[nt] = w.conv2_b[col]; lw[nt] = w.lin_w[col];
            g2[nt] = q == 2 ? w.ln2_g[col] : 0.0f; be2[nt] = q == 2 ? w.ln2_b[col] : 0.0f;
        }
        const float lb = w.lin_b[0];
        // bucket edges replicated in both half waves: lane l holds edges (l&31) + 32*e; +inf beyond the dim-1 edges
        float edge[ND];
#pragma unroll
        for (int e = 0; e < ND; ++e) edge[e] = (q < 2 && 32 * e + i < DIM - 1) ? w.bins[32 * e + i] : INFINITY;
        f32x16 c[ND];
        zero_tiles<ND>(c);
        {
            const float* const taps[3] = {f_row - LDD, f_row, f_row + LDD};
            WaveGrp<ND> gw;
            wave_prefetch<ND>(gw, w.conv1_w, ND, 0, 0, lane);
            wave_gemm_taps<ND, 3, ND, false>(c, gw, taps, tok3, 3, w.conv1_w, (long)DIM * DIM, ND, 0, 0, lane);
        }
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
        __syncthreads();   // previous predictor finished reading tb
        tile_store<ND>(tb, LDD, 0, c, lane);
        __syncthreads();
        zero_tiles<ND>(c);
        {
            const float* const taps[3] = {t_row - LDD, t_row, t_row + LDD};
            WaveGrp<ND> gw;
            wave_prefetch<ND>(gw, w.conv2_w, ND, 0, 0, lane);
            wave_gemm_taps<ND, 3, ND, false>(c, gw, taps, tok3, 3, w.conv2_w, (long)DIM * DIM, ND, 0, 0, lane);
        }
        ESMI_CT();   // conv2 done
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c[nt][r] = fmaxf(c[nt][r] + b2[nt], 0.0f);
        }
        float pr[16];      // Linear(dim, 1) on the pre-norm2 tensor (networks.py:157-160)
        int bidx[16];      // torch.bucketize(v, edges, right=False) = number of edges strictly below v
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
                const float* tv = q == 0 ? p.pitch_t : p.energy_t;
                const float v = (tv && !rout[r]) ? tv[(long)b * p.T + rpos[r]] : s;
#pragma unroll
                for (int e = 0; e < ND; ++e) {
                    const unsigned long long m = ballot64(edge[e] < v);
                    bidx[r] += __builtin_popcount((unsigned)(h2 ? (m >> 32) : m));
                }
            }
        }
        ESMI_CT();   // dot done
        if (q == 2) layernorm_tile_regs<ND>(c, g2, be2);   // duration features (networks.py:161-163)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tile_row(r, lane);
            const bool live = row >= 2 && row < 2 + kVaTileRows && !rout[r];
            const long grow = live ? (long)b * p.T + rpos[r] : 0;
            float* frow = p.feat + grow * 4 * DIM + (q == 2 ? 3 : 1 + q) * DIM + i;
            if (live && i == 0) p.preds[q][grow] = pr[r];
            if (q == 2) {
#pragma unroll
                for (int nt = 0; nt < ND; ++nt)
                    if (live) frow[32 * nt] = rz[r] ? 0.0f : c[nt][r];
                if (live && i == 0) {
                    float d = p.dur_t ? (float)p.dur_t[grow] : rintf(pr[r]);   // torch.round: half to even
                    if (p.mask) {                                              // networks.py:381-382
                        if (rz[r]) d = 0.0f;
                        d = fmaxf(d, 0.0f);
                    }
                    p.dur[grow] = (int)d;
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < ND; ++nt)
                    if (live) frow[32 * nt] = rz[r] ? 0.0f : w.emb[bidx[r] * DIM + 32 * nt + i];
                if (live && i == 0) {
                    int* ip = q == 0 ? p.pitch_idx : p.energy_idx;
                    if (ip) ip[grow] = bidx[r];
                }
            }
        }
        ESMI_CT();   // outputs done
    }
}

}  // namespace esmi
