// Fused second half of an encoder block (layers/networks.py:72-85, layers/blocks.py:22-29,49-66):
//
//     y  = proj(softmax(q k^T * scale) v) + bias            SelfAttention (scores NOT masked, heads full width)
//     y1 = mask(LN1(y + x))                                  networks.py:73-75
//     f  = mask(LN2(mlp2(GELU(conv3(mlp1(y1)))) + y1))       MixFFN + networks.py:80-83
//
// A workgroup of nw <= 4 waves owns 32*nw consecutive positions of one utterance, one 32-row MFMA tile per wave.
// Everything is row-local except the dense k=3 conv inside MixFFN, which needs the neighbouring rows' mlp1
// outputs: the waves of a workgroup exchange those through one shared LDS tile [32*nw + 2][hidden + 4] (two
// workgroup barriers); only at a workgroup edge that is not a sequence edge is a halo row recomputed (attention
// included).  When the whole sequence fits (N <= 128) nothing is recomputed: tiny ES, T = 128 -> exactly one wave
// per SIMD (the earlier one-wave-per-30-row-tile version launched 1280 waves = two rounds on the 1024 SIMDs).
//
// Memory discipline (it decides the speed of these latency-bound kernels): every load and store is unconditional
// in the instruction stream -- ragged edges go through bounds-checked buffer accesses -- and the first weight group
// of every GEMM is requested one stage ahead (wave_prefetch), so hipcc can count vmcnt exactly and no stage starts
// with an exposed round trip.  Six kernel launches and five HBM round trips of the unfused path become one launch.
#pragma once
#include "enc_merge_qkv.h"
#include "wave_chain.h"

namespace esmi {

struct EncAttnFfnP {
    const float* x;    // (B,N,C)   block input after the merge convs (first residual)
    const float* qkv;  // (B,N,3,h,C)
    int B, N, C, h;
    float scale;
    // the four weight matrices are in MFMA B-fragment order (esmi_pack_bfrag_f32, see wave_chain.h)
    const float *proj_w, *proj_b;   // (C, h*C), (C)
    const float *ln1_g, *ln1_b;
    // MixFFN's Linear(C, E*C) folded into its dense k = 3 conv (esmi.h, ffn_cw): one k = 3 conv C -> E*C.  ffn_b = the bias of an interior
    // row; a row whose first / last tap falls off the sequence lacks that tap's share of the Linear's bias: ffn_b0 / ffn_b2
    const float *ffn_w;             // (3, E*C, C) tap-major, MFMA B-fragment order
    const float *ffn_b, *ffn_b0, *ffn_b2;   // (E*C) each
    const float *mlp2_w, *mlp2_b;   // (C, E*C)
    const float *ln2_g, *ln2_b;
    const unsigned char* mask;      // (B, mask_len) or NULL; row n is padding iff any of mask[n*mask_pool .. +mask_pool) is set / beyond mask_len
    int mask_pool, mask_len;
    float* out;                     // (B,N,C)
    int wgs_per_b;                  // workgroups per utterance
    int useful;                     // positions stored per workgroup: 32*nw - 2*halo
    int halo;                       // 0: one workgroup covers the sequence, 1: one recomputed row per side
    EncMergeP m;                    // whole-block instantiations (NCI > 0) only: the merge conv / qkv stage's inputs
    int fold;                       // whole-block instantiations only: 1 = weight-folded attention (esmi.h, qk_w / vo_w): m.qkv_w is the (h*C, C)
                                    // matrix of q_h = x M_h, keys = values = x for every head, proj_w is the (C, h*C) output matrix
};

constexpr int kEncMaxWaves = 4;     // waves per workgroup (one per SIMD)

// workgroup shape for a sequence of n positions: fewest waves in total, ties to the larger workgroup
inline void enc_attn_ffn_plan(int n, int ld_floats, int* nw, int* wgs, int* useful, int* halo) {
    int nwmax = kEncMaxWaves;
    while (nwmax > 1 && (32 * nwmax + 2) * ld_floats * 4 > 150 * 1024) --nwmax;
    if (n <= 32 * nwmax) { *nw = (n + 31) / 32; *wgs = 1; *useful = 32 * *nw; *halo = 0; return; }
    int best = 1, best_waves = 1 << 30;
    for (int w = 1; w <= nwmax; ++w) {
        const int u = 32 * w - 2, waves = ((n + u - 1) / u) * w;
        if (waves <= best_waves) { best_waves = waves; best = w; }
    }
    *nw = best; *useful = 32 * best - 2; *wgs = (n + *useful - 1) / *useful; *halo = 1;
}

// row stride of the LDS-resident qkv tile of the whole-block kernels: == 4 (mod 64) floats, so that the 16-byte
// K / Q row fragments of 32 consecutive rows are bank-conflict free
__host__ __device__ inline int enc_qkv_ld(int h, int C) { return ((3 * h * C + 63) & ~63) + 4; }
inline int enc_block_lds_floats(int C, int h, int expansion, int c_in, int k, int stride, int nw) {
    const int stg = nw * (31 * stride + k) * (c_in + 4), qkv = 32 * nw * enc_qkv_ld(h, C);
    return (32 * nw + 2) * (expansion * C + 4) + (stg > qkv ? stg : qkv);
}

// Attention contractions of the chain kernels (both operands are activations).  Split build: the operands are split into two
// binary16 pieces on the fly and the three significant products run as v_mfma_f32_32x32x16_f16 (3 x 32 cycles per 16 channels /
// keys and tile instead of 8 x 64 for v_mfma_f32_32x32x2_f32; the same 22-bit products as every weight GEMM and as
// attn_lds_kernel); the exact-fp32 build keeps the fp32 instruction.  Block-0 trace before: S^T 8.5k + P V 8.7k cycles of 54k.
//   s[kt] += K(32 keys x 32 channels) Q^T: qv / kv = the lane's four 16-byte fragments (channels [8g + 4h, +4), g < 4)
template <int NKT>
__device__ __forceinline__ void attn_scores_grp(f32x16 (&s)[NKT], const f32x4 (&qv)[4], const f32x4 (&kv)[4][NKT]) {
#if ESMI_CHAIN_SPLIT
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const f16x2p qf = split_f16x2(qv[2 * st], qv[2 * st + 1]);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const f16x2p kf = split_f16x2(kv[2 * st][kt], kv[2 * st + 1][kt]);
            s[kt] = mfma32_f16(kf.h2, qf.h1, s[kt]);
            s[kt] = mfma32_f16(kf.h1, qf.h2, s[kt]);
            s[kt] = mfma32_f16(kf.h1, qf.h1, s[kt]);
        }
    }
#else
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) s[kt] = mfma32(kv[g][kt][t], qv[g][t], s[kt]);
        }
    }
#endif
}
//   o[nt] += P(32 queries x 16 keys) V: pa / pb = the lane's P values of accumulator registers [8u, +4) / [8u + 4, +4) of a key
//   tile (keys 16u + {0..3} + 4h and 16u + 8 + {0..3} + 4h: the half waves' eight k-slots of one 16-key step), va / vb = V of
//   those keys at this lane's column of every column tile
template <int NC>
__device__ __forceinline__ void attn_pv_step(f32x16 (&o)[NC], const f32x4& pa, const f32x4& pb, const float (&va)[4][NC],
                                             const float (&vb)[4][NC]) {
#if ESMI_CHAIN_SPLIT
    const f16x2p pf = split_f16x2(pa, pb);
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) {
        const f32x4 a = {va[0][nt], va[1][nt], va[2][nt], va[3][nt]}, b = {vb[0][nt], vb[1][nt], vb[2][nt], vb[3][nt]};
        const f16x2p vf = split_f16x2(a, b);
        o[nt] = mfma32_f16(pf.h2, vf.h1, o[nt]);
        o[nt] = mfma32_f16(pf.h1, vf.h2, o[nt]);
        o[nt] = mfma32_f16(pf.h1, vf.h1, o[nt]);
    }
#else
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
        for (int nt = 0; nt < NC; ++nt) o[nt] = mfma32(pa[rr], va[rr][nt], o[nt]);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
        for (int nt = 0; nt < NC; ++nt) o[nt] = mfma32(pb[rr], vb[rr][nt], o[nt]);
    }
#endif
}

// NCI == 0: x and qkv come from global memory (written by enc_merge_qkv_kernel).
// NCI  > 0: WHOLE BLOCK in one launch, for sequences one workgroup covers (N <= 128): each wave first runs the merge
//           conv + qkv stage of its 32 rows (Cin = 32*NCI, kernel KT, stride STRIDE); q/k/v go to an LDS tile shared by
//           the workgroup and x stays in registers as the residual -- neither ever touches HBM.
template <int NKT, int NC, int E, int NCI = 0, int KT = 1, int STRIDE = 1>   // keys <= 32*NKT, C = 32*NC, MixFFN hidden = E*C
__device__ __forceinline__ void enc_attn_ffn_body(const EncAttnFfnP& p) {
    constexpr int NE = NC * E;
    constexpr int C = 32 * NC, EC = 32 * NE;
    constexpr int LD = EC + 4;
    constexpr bool FUSED = NCI > 0;
    ESMI_DYN_LDS(lds);              // [32*nw + 2][LD]: first and last row are the zero rows around the workgroup's tile
    ESMI_CT_INIT(NC == 1 ? 0 : 1);
    ESMI_CT();   // entry
    const int nw = (int)(blockDim.x >> 6), w = wave_id();
    const int lane = lane_id(), i = lane & 31, h2 = lane >> 5;
    const int b = (int)blockIdx.x / p.wgs_per_b, wg = (int)blockIdx.x - b * p.wgs_per_b;
    const int r0 = 32 * w;                                // workgroup-local row of this wave's row 0
    const int t0 = wg * p.useful - p.halo + r0;            // sequence position of this wave's row 0
    float* buf = lds + LD * (1 + r0);                      // this wave's 32 rows
    for (int c = (int)threadIdx.x; c < LD; c += (int)blockDim.x) {
        lds[c] = 0.0f;
        lds[(32 * nw + 1) * LD + c] = 0.0f;
    }
    const int pos_i = t0 + i;
    const float* a_row = buf + i * LD + 4 * h2;
    const int ld = 3 * p.h * C;
    const BufRsrc r_qkv = make_rsrc(p.qkv + (long)b * p.N * ld, (long)p.N * ld * 4);
    const BufRsrc r_x = make_rsrc(p.x + (long)b * p.N * C, (long)p.N * C * 4);
    const BufRsrc r_mask = make_rsrc(p.mask ? p.mask + (long)b * p.mask_len : nullptr, p.mask_len);
    const BufRsrc r_out = make_rsrc(p.out + (long)b * p.N * C, (long)p.N * C * 4);
    ESMI_CT();   // 0 start
    const int ldq = enc_qkv_ld(p.h, C);
    float* qkv_t = lds + (32 * nw + 2) * LD;               // FUSED: [32*nw][ldq], aliased by the waves' input staging tiles
    f32x16 xacc[FUSED ? NC : 1];
    if constexpr (FUSED) {
        // ---------------- merge conv + qkv of this wave's rows (enc_merge_qkv.h), results stay on the CU
        constexpr int CIN = 32 * NCI;
        float* stg = qkv_t + w * ((31 * STRIDE + KT) * (CIN + 4));
        merge_conv_tile<NCI, NC, KT, STRIDE>(p.m, b, r0, stg, lane, xacc);
        ESMI_CT();   // merge conv done
        if (p.fold) {
            // weight-folded attention: only q_h = x M_h is a contraction (h*C columns instead of 3*h*C); keys = values = x for every
            // head: one copy of this wave's x rows behind the q columns of the shared tile
            const int ntq = (p.h * C) >> 5;
            WaveGrp<NC> gq;
            wave_prefetch<NC>(gq, p.m.qkv_w, ntq, 0, 0, lane);
            tile_store<NC>(buf, LD, 0, xacc, lane);
            __syncthreads();        // every wave is done with its staging tile: the region becomes the q | x tile
#pragma unroll
            for (int nt = 0; nt < NC; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) qkv_t[(r0 + tile_row(r, lane)) * ldq + p.h * C + 32 * nt + i] = xacc[nt][r];
            }
            for (int hd = 0; hd < p.h; ++hd) {
                f32x16 q[NC];
                zero_tiles<NC>(q);
                wave_gemm_k<NC, NC>(q, gq, a_row, true, p.m.qkv_w, ntq, 0, hd * NC, lane);
                if (hd + 1 < p.h) wave_prefetch<NC>(gq, p.m.qkv_w, ntq, 0, (hd + 1) * NC, lane);
#pragma unroll
                for (int nt = 0; nt < NC; ++nt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) qkv_t[(r0 + tile_row(r, lane)) * ldq + hd * C + 32 * nt + i] = q[nt][r];
                }
            }
        } else {
        const int nq = 3 * p.h * C, ntq = nq >> 5;
        WaveGrp<4> gq;
        wave_prefetch<4>(gq, p.m.qkv_w, ntq, 0, 0, lane);
        tile_store<NC>(buf, LD, 0, xacc, lane);
        __syncthreads();            // every wave is done with its staging tile: the region becomes the qkv tile
        for (int n0 = 0; n0 < nq; n0 += 128) {
            f32x16 q[4];
            zero_tiles<4>(q);
            wave_gemm_k<4, NC>(q, gq, a_row, true, p.m.qkv_w, ntq, 0, n0 >> 5, lane);
            if (n0 + 128 < nq) wave_prefetch<4>(gq, p.m.qkv_w, ntq, 0, (n0 + 128) >> 5, lane);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                if (n0 + 32 * nt < nq) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) qkv_t[(r0 + tile_row(r, lane)) * ldq + n0 + 32 * nt + i] = q[nt][r];
                }
            }
        }
        }
        __syncthreads();            // q / k / v of the whole sequence are in place
        ESMI_CT();   // qkv done
    }

    // ---------------- prologue: everything that does not depend on a result is requested now, nothing is waited for.
    // Rows outside [0, N) are out of range of their buffers: negative positions wrap to huge unsigned offsets.
    unsigned mb = 0;                // blocks.py:51-57: the mask is padded with True and max-pooled by the block's total stride
    for (int q = 0; q < p.mask_pool; ++q) {
        const int idx = pos_i * p.mask_pool + q;
        mb |= buf_ld_u8(r_mask, pos_i >= 0 ? (unsigned)idx : kBufOOB) | (unsigned)(p.mask && pos_i >= 0 && idx >= p.mask_len);
    }
    const unsigned q_off = (unsigned)((pos_i * ld + 4 * h2) * 4);
    unsigned k_off[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) k_off[kt] = (unsigned)((((32 * kt + i) * ld) + p.h * C + 4 * h2) * 4);
    WaveGrp<NC> gp;                 // proj weights, head 0
    wave_prefetch<NC>(gp, p.proj_w, NC, 0, 0, lane);
    constexpr bool kHoistRes = NC <= 2 && !FUSED;
    f32x16 xres[kHoistRes ? NC : 1];
    if (kHoistRes) {
#pragma unroll
        for (int nt = 0; nt < (kHoistRes ? NC : 1); ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                xres[nt][r] = buf_ld(r_x, (unsigned)(((t0 + tile_row(r, lane)) * C + 32 * nt + i) * 4));
        }
    }
    float pb_[NC], g1_[NC], be1_[NC], b2_[NC], g2_[NC], be2_[NC], cb_[NE], cb0_[NE], cb2_[NE];
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) {
        const int col = 32 * nt + i;
        pb_[nt] = p.proj_b[col]; g1_[nt] = p.ln1_g[col]; be1_[nt] = p.ln1_b[col];
        b2_[nt] = p.mlp2_b[col]; g2_[nt] = p.ln2_g[col]; be2_[nt] = p.ln2_b[col];
    }
#pragma unroll
    for (int nt = 0; nt < NE; ++nt) {
        cb_[nt] = p.ffn_b[32 * nt + i];
        cb0_[nt] = p.ffn_b0[32 * nt + i];
        cb2_[nt] = p.ffn_b2[32 * nt + i];
    }

    // ---------------- attention, one head at a time; proj accumulates over heads
    f32x16 y[NC];
    zero_tiles<NC>(y);
    for (int hd = 0; hd < p.h; ++hd) {
        const unsigned hd_off = (unsigned)(hd * C * 4);
        const unsigned v_base = (unsigned)(((2 * p.h + hd) * C + i) * 4);
        f32x16 s[NKT];
        zero_tiles<NKT>(s);
        for (int kc = 0; kc < (C >> 3); kc += 4) {   // S^T[key][query] = sum_c K[key][c] Q[query][c]
            f32x4 qv[4], kv[4][NKT];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (FUSED) {
                    qv[g] = ld4(qkv_t + (r0 + i) * ldq + hd * C + 8 * (kc + g) + 4 * h2);
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) {   // key tiles beyond the workgroup's rows alias its last tile (their scores are masked)
                        const int krow = 32 * (kt < nw ? kt : nw - 1) + i;
                        kv[g][kt] = ld4(qkv_t + krow * ldq + (p.fold ? p.h : p.h + hd) * C + 8 * (kc + g) + 4 * h2);
                    }
                } else {
                    qv[g] = buf_ld4(r_qkv, q_off + hd_off + 32u * (kc + g));
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) kv[g][kt] = buf_ld4(r_qkv, k_off[kt] + hd_off + 32u * (kc + g));
                }
            }
            attn_scores_grp<NKT>(s, qv, kv);
        }
        // P V operands: 4*NKT groups of four key rows; the first two groups are requested before the softmax.
        // keys >= N: V reads 0 (buffer bound) and P = 0.
        struct VG { float v[4][NC]; };
        auto vfetch = [&](int f, VG& gq) __attribute__((always_inline)) {
            const int kt = f >> 2, r4 = (f & 3) << 2;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int key = 32 * kt + tile_row(r4 + rr, lane);   // differs between the half waves: that IS the k index
#pragma unroll
                for (int nt = 0; nt < NC; ++nt) {
                    if constexpr (FUSED)   // keys beyond the workgroup's rows: P = 0, any finite V will do
                        gq.v[rr][nt] = qkv_t[(key < 32 * nw ? key : 32 * nw - 1) * ldq + (p.fold ? p.h : 2 * p.h + hd) * C + 32 * nt + i];
                    else gq.v[rr][nt] = buf_ld(r_qkv, v_base + (unsigned)((key * ld + 32 * nt) * 4));
                }
            }
        };
        VG v0, v1;
        vfetch(0, v0);
        vfetch(1, v1);
        ESMI_CT();   // 1 S^T issued
        float mx = -INFINITY;   // softmax over keys of this lane's query: in-lane, then the other half wave
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * kt + tile_row(r, lane);
                const float v = key < p.N ? s[kt][r] * p.scale : -INFINITY;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = fmaxf(mx, swap32_f(mx));
        float den = 0.0f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = exp_fast_f32(s[kt][r] - mx);
                s[kt][r] = e;
                den += e;
            }
        }
        den += swap32_f(den);
        const float inv = rcp_fast_f32(den);
        ESMI_CT();   // 2 softmax done
        f32x16 o[NC];           // ctx[query][c] = sum_key P[query][key] V[key][c]
        zero_tiles<NC>(o);
#pragma unroll
        for (int f = 0; f < 4 * NKT; f += 2) {   // one 16-key step: accumulator registers [4f, 4f + 8) of key tile f >> 2
            f32x4 pa, pb;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pa[e] = s[f >> 2][((f & 3) << 2) + e] * inv;
                pb[e] = s[f >> 2][((f & 3) << 2) + 4 + e] * inv;
            }
            VG w0 = v0, w1 = v1;
            if (f + 2 < 4 * NKT) vfetch(f + 2, v0);
            if (f + 3 < 4 * NKT) vfetch(f + 3, v1);
            attn_pv_step<NC>(o, pa, pb, w0.v, w1.v);
        }
        ESMI_CT();   // 3 PV issued
        lds_wave_sync();        // the previous head's proj has finished reading the tile
        tile_store<NC>(buf, LD, 0, o, lane);
        lds_wave_sync();
        {   // this head's slice of the projection, added to the heads before it (a GEMM starts from zero: wave_acc_join)
            f32x16 yh[NC];
            zero_tiles<NC>(yh);
            wave_gemm_k<NC, NC>(yh, gp, a_row, true, p.proj_w, NC, (hd * C) >> 3, 0, lane);
#pragma unroll
            for (int nt = 0; nt < NC; ++nt) y[nt] += yh[nt];
        }
        if (hd + 1 < p.h) wave_prefetch<NC>(gp, p.proj_w, NC, ((hd + 1) * C) >> 3, 0, lane);
    }
    WaveGrp<NE> gm;                 // MixFFN conv weights, tap 0
    wave_prefetch<NE>(gm, p.ffn_w, NE, 0, 0, lane);

    // rows that are padding (mask) / outside the sequence
    const unsigned mbits = (unsigned)ballot64(mb != 0);   // bit i = row i (both half waves hold the same rows)
    bool rz[16], rout[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = tile_row(r, lane);
        const int pos = t0 + row;
        rout[r] = pos < 0 || pos >= p.N;
        rz[r] = !rout[r] && ((mbits >> row) & 1u);
    }
    ESMI_CT();   // 4 proj issued
    // ---------------- y1 = mask(LN1(y + bias + x))
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) {
        const int col = 32 * nt + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float xr;
            if (FUSED) xr = xacc[FUSED ? nt : 0][r];
            else if (kHoistRes) xr = xres[kHoistRes ? nt : 0][r];
            else xr = buf_ld(r_x, (unsigned)(((t0 + tile_row(r, lane)) * C + col) * 4));
            y[nt][r] += pb_[nt] + xr;
        }
    }
    layernorm_tile_regs<NC>(y, g1_, be1_);
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (rz[r] || rout[r]) y[nt][r] = 0.0f;   // (rows outside the sequence: the MixFFN conv's zero padding; never stored)
    }
    lds_wave_sync();
    tile_store<NC>(buf, LD, 0, y, lane);
    __syncthreads();            // the neighbouring waves' boundary rows (and the zero rows) are in place

    ESMI_CT();   // 5 LN1 + store done
    // ---------------- MixFFN: (Linear folded into) dense conv k3 -> GELU -> mlp2
    f32x16 m[NE];
    ESMI_CT();   // 6 (the stage of the separate Linear, gone)
    zero_tiles<NE>(m);
    {
        const float* const taps[3] = {a_row - LD, a_row, a_row + LD};
        const bool tok[3] = {true, true, true};
        wave_gemm_taps<NE, 3, NC, false>(m, gm, taps, tok, p.ffn_w, (long)EC * C, NE, 0, 0, lane);
    }
    WaveGrp<NC> g2;                 // mlp2 weights
    wave_prefetch<NC>(g2, p.mlp2_w, NC, 0, 0, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int pos = t0 + tile_row(r, lane);
#pragma unroll
        for (int nt = 0; nt < NE; ++nt) {
            const float bias = cb_[nt] - (pos == 0 ? cb0_[nt] : 0.0f) - (pos == p.N - 1 ? cb2_[nt] : 0.0f);
            m[nt][r] = gelu_fast_f32(m[nt][r] + bias);
        }
    }
    __syncthreads();            // every wave has read its neighbours' rows
    tile_store<NE>(buf, LD, 0, m, lane);
    lds_wave_sync();
    ESMI_CT();   // 7 conv + gelu + store
    f32x16 z[NC];
    zero_tiles<NC>(z);
    wave_gemm_k<NC, NE>(z, g2, a_row, true, p.mlp2_w, NC, 0, 0, lane);
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) z[nt][r] += b2_[nt] + y[nt][r];
    }
    ESMI_CT();   // 8 mlp2
    layernorm_tile_regs<NC>(z, g2_, be2_);
    ESMI_CT();   // 9 LN2
    const int row_lo = p.halo, row_hi = 32 * nw - p.halo;   // workgroup-local rows this workgroup stores
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = tile_row(r, lane);
        const bool keep = r0 + row >= row_lo && r0 + row < row_hi && t0 + row >= 0;   // rows >= N fall off the buffer end
        const unsigned off = keep ? (unsigned)(((t0 + row) * C + i) * 4) : kBufOOB;
#pragma unroll
        for (int nt = 0; nt < NC; ++nt) buf_st(r_out, off + 128u * nt, rz[r] ? 0.0f : z[nt][r]);
    }
}

template <int NKT, int NC, int E, int NCI = 0, int KT = 1, int STRIDE = 1>
__global__ __launch_bounds__(64 * kEncMaxWaves, kChainWps) void enc_attn_ffn_kernel(const EncAttnFfnP p) {
    enc_attn_ffn_body<NKT, NC, E, NCI, KT, STRIDE>(p);
}


}  // namespace esmi

// ---------------------------------------------------------------------------------------------------------------
// Column-split variant for two-head blocks on short sequences (tiny ES block 1: N = 64, C = 64, h = 2).  With one
// wave per 32 rows such a block keeps only B*N/32 waves busy (512 of the 1024 SIMDs at B = 256) on a long serial
// chain.  Here TWO waves share a row tile: wave c computes the attention of head c, then owns half of the output
// columns of every later GEMM (proj over both heads' contexts, mlp1, conv, mlp2).  What crosses between the two goes
// through the shared LDS tile (contexts, y1, hidden) or, for the two LayerNorms, as per-row (mean, M2) pairs that are
// merged with the parallel-variance formula -- both waves compute bit-identical statistics.
namespace esmi {

constexpr int kEncSplitMaxTiles = 2;     // row tiles per workgroup (2 waves each)

inline void enc_attn_ffn_split_plan(int n, int* nw, int* wgs, int* useful, int* halo) {
    if (n <= 32 * kEncSplitMaxTiles) { *nw = (n + 31) / 32; *wgs = 1; *useful = 32 * *nw; *halo = 0; return; }
    *nw = kEncSplitMaxTiles; *useful = 32 * *nw - 2; *wgs = (n + *useful - 1) / *useful; *halo = 1;
}
inline int enc_attn_ffn_split_lds_floats(int C, int h, int expansion, int nw) {
    const int wide = (h * C > expansion * C ? h * C : expansion * C) + 4;
    return (32 * nw + 2) * wide + nw * 2 * 32 * 2;
}

// LayerNorm over C = 2 * 32*NH columns of which this wave holds 32*NH; partner statistics through `stats`
// ([tile][wave][32 rows][2]); two workgroup barriers inside.
template <int NH>
__device__ __forceinline__ void layernorm_split(f32x16 (&v)[NH], const float (&gg)[NH], const float (&bb)[NH], float* stats, int rt,
                                                int c, int lane, float eps = 1e-5f) {
    const int i = lane & 31;
    const float inv_h = 1.0f / (float)(32 * NH);
    float mean_l[16], m2_l[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        mean_l[r] = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NH; ++nt) mean_l[r] += v[nt][r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) mean_l[r] = row_sum32(mean_l[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        mean_l[r] *= inv_h;
        m2_l[r] = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NH; ++nt) {
            const float d = v[nt][r] - mean_l[r];
            m2_l[r] = fmaf(d, d, m2_l[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) m2_l[r] = row_sum32(m2_l[r]);
    __syncthreads();            // the statistics buffer is free (previous LayerNorm fully consumed)
    float* mine = stats + ((rt * 2 + c) * 32) * 2;
    const float* other = stats + ((rt * 2 + (c ^ 1)) * 32) * 2;
    if (i == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            mine[2 * tile_row(r, lane)] = mean_l[r];
            mine[2 * tile_row(r, lane) + 1] = m2_l[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float mo = other[2 * tile_row(r, lane)], qo = other[2 * tile_row(r, lane) + 1];
        const float ma = c == 0 ? mean_l[r] : mo, mb = c == 0 ? mo : mean_l[r];   // same operand order in both waves
        const float qa = c == 0 ? m2_l[r] : qo, qb = c == 0 ? qo : m2_l[r];
        const float mean = 0.5f * (ma + mb);
        const float d = mb - ma;
        const float m2 = (qa + qb) + d * d * (float)(16 * NH);                      // n_a n_b / (n_a + n_b) = 32*NH / 2
        const float rstd = rsqrt_fast_f32(m2 * (0.5f * inv_h) + eps);
#pragma unroll
        for (int nt = 0; nt < NH; ++nt) v[nt][r] = fmaf((v[nt][r] - mean) * rstd, gg[nt], bb[nt]);
    }
}

inline int enc_block_split_lds_floats(int C, int h, int expansion, int c_in, int k, int stride, int nw) {
    const int stg = 2 * nw * (31 * stride + k) * (c_in + 4), qkv = 32 * nw * enc_qkv_ld(h, C);
    return enc_attn_ffn_split_lds_floats(C, h, expansion, nw) + (stg > qkv ? stg : qkv);
}

// NCI > 0: whole block in one launch (see enc_attn_ffn_kernel): both waves of a pair run the (cheap) merge conv of their
// row tile, wave c then computes the q / k / v columns of head c into the shared LDS tile.
template <int NKT, int NC, int E, int NCI = 0, int KT = 1, int STRIDE = 1>   // h == 2; keys <= 32*NKT, C = 32*NC (NC even), MixFFN hidden = E*C
__device__ __forceinline__ void enc_attn_ffn_split_body(const EncAttnFfnP& p) {
    constexpr int NE = NC * E, NCH = NC / 2, NEH = NE / 2;
    constexpr int C = 32 * NC, EC = 32 * NE, HC = 2 * C;
    constexpr int LD = (HC > EC ? HC : EC) + 4;
    constexpr bool FUSED = NCI > 0;
    ESMI_DYN_LDS(lds);              // [32*nw + 2][LD] shared tile (zero rows around), then the LayerNorm statistics
    ESMI_CT_INIT(1);
    ESMI_CT();   // entry
    const int nw = (int)(blockDim.x >> 7), w = wave_id();
    const int rt = w >> 1, c = w & 1;                       // row tile, column half / head
    const int lane = lane_id(), i = lane & 31, h2 = lane >> 5;
    const int b = (int)blockIdx.x / p.wgs_per_b, wg = (int)blockIdx.x - b * p.wgs_per_b;
    const int r0 = 32 * rt;
    const int t0 = wg * p.useful - p.halo + r0;
    float* buf = lds + LD * (1 + r0);
    float* stats = lds + (32 * nw + 2) * LD;
    float* qkv_t = stats + nw * 2 * 32 * 2;                 // FUSED: [32*nw][ldq], aliased by the waves' input staging tiles
    for (int cc = (int)threadIdx.x; cc < LD; cc += (int)blockDim.x) {
        lds[cc] = 0.0f;
        lds[(32 * nw + 1) * LD + cc] = 0.0f;
    }
    const int pos_i = t0 + i;
    const float* a_row = buf + i * LD + 4 * h2;
    const int ld = 3 * 2 * C;
    const int ldq = enc_qkv_ld(2, C);
    const BufRsrc r_qkv = make_rsrc(p.qkv + (long)b * p.N * ld, (long)p.N * ld * 4);
    const BufRsrc r_x = make_rsrc(p.x + (long)b * p.N * C, (long)p.N * C * 4);
    const BufRsrc r_mask = make_rsrc(p.mask ? p.mask + (long)b * p.mask_len : nullptr, p.mask_len);
    const BufRsrc r_out = make_rsrc(p.out + (long)b * p.N * C, (long)p.N * C * 4);
    const int c0 = 32 * NCH * c;                            // first output column of this wave (width C stages)
    const int e0 = 32 * NEH * c;                            // first hidden column of this wave
    f32x16 xacc[FUSED ? NC : 1];
    if constexpr (FUSED) {
        constexpr int CIN = 32 * NCI;
        float* stg = qkv_t + w * ((31 * STRIDE + KT) * (CIN + 4));
        merge_conv_tile<NCI, NC, KT, STRIDE>(p.m, b, r0, stg, lane, xacc);
        ESMI_CT();   // merge conv done
        const int ntq = ((p.fold ? 1 : 3) * 2 * C) >> 5;
        WaveGrp<NC> gq;
        wave_prefetch<NC>(gq, p.m.qkv_w, ntq, 0, c * NC, lane);
        if (c == 0) tile_store<NC>(buf, LD, 0, xacc, lane);   // both waves of the pair hold the same x
        __syncthreads();            // x tile written; every wave is done with its staging tile
        if (p.fold) {               // weight-folded attention (see enc_attn_ffn_body): q of head c, and this wave's half of the x copy
#pragma unroll
            for (int nt = 0; nt < NCH; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    qkv_t[(r0 + tile_row(r, lane)) * ldq + 2 * C + c0 + 32 * nt + i] = c ? xacc[FUSED ? NCH + nt : 0][r] : xacc[FUSED ? nt : 0][r];
            }
            f32x16 q[NC];
            zero_tiles<NC>(q);
            wave_gemm_k<NC, NC>(q, gq, a_row, true, p.m.qkv_w, ntq, 0, c * NC, lane);
#pragma unroll
            for (int nt = 0; nt < NC; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) qkv_t[(r0 + tile_row(r, lane)) * ldq + c * C + 32 * nt + i] = q[nt][r];
            }
        } else {
#pragma unroll
        for (int part = 0; part < 3; ++part) {                // q, k, v columns of head c
            f32x16 q[NC];
            zero_tiles<NC>(q);
            wave_gemm_k<NC, NC>(q, gq, a_row, true, p.m.qkv_w, ntq, 0, (2 * part + c) * NC, lane);
            if (part < 2) wave_prefetch<NC>(gq, p.m.qkv_w, ntq, 0, (2 * (part + 1) + c) * NC, lane);
#pragma unroll
            for (int nt = 0; nt < NC; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    qkv_t[(r0 + tile_row(r, lane)) * ldq + (2 * part + c) * C + 32 * nt + i] = q[nt][r];
            }
        }
        }
        __syncthreads();            // q / k / v of the whole sequence are in place; the x tile has been read
        ESMI_CT();   // qkv done
    }

    // ---------------- prologue loads (nothing waited for)
    unsigned mb = 0;
    for (int q = 0; q < p.mask_pool; ++q) {
        const int idx = pos_i * p.mask_pool + q;
        mb |= buf_ld_u8(r_mask, pos_i >= 0 ? (unsigned)idx : kBufOOB) | (unsigned)(p.mask && pos_i >= 0 && idx >= p.mask_len);
    }
    const unsigned q_off = (unsigned)((pos_i * ld + c * C + 4 * h2) * 4);
    unsigned k_off[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) k_off[kt] = (unsigned)((((32 * kt + i) * ld) + (2 + c) * C + 4 * h2) * 4);
    const unsigned v_base = (unsigned)(((4 + c) * C + i) * 4);
    WaveGrp<NCH> gp;
    wave_prefetch<NCH>(gp, p.proj_w, NC, 0, c * NCH, lane);
    f32x16 xres[NCH];
#pragma unroll
    for (int nt = 0; nt < NCH; ++nt) {
        if constexpr (FUSED) {      // this wave's column half of x (static register indices, wave-uniform select)
#pragma unroll
            for (int r = 0; r < 16; ++r) xres[nt][r] = c ? xacc[FUSED ? NCH + nt : 0][r] : xacc[FUSED ? nt : 0][r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                xres[nt][r] = buf_ld(r_x, (unsigned)(((t0 + tile_row(r, lane)) * C + c0 + 32 * nt + i) * 4));
        }
    }
    float pb_[NCH], g1_[NCH], be1_[NCH], b2_[NCH], g2_[NCH], be2_[NCH], cb_[NEH], cb0_[NEH], cb2_[NEH];
#pragma unroll
    for (int nt = 0; nt < NCH; ++nt) {
        const int col = c0 + 32 * nt + i;
        pb_[nt] = p.proj_b[col]; g1_[nt] = p.ln1_g[col]; be1_[nt] = p.ln1_b[col];
        b2_[nt] = p.mlp2_b[col]; g2_[nt] = p.ln2_g[col]; be2_[nt] = p.ln2_b[col];
    }
#pragma unroll
    for (int nt = 0; nt < NEH; ++nt) {
        cb_[nt] = p.ffn_b[e0 + 32 * nt + i];
        cb0_[nt] = p.ffn_b0[e0 + 32 * nt + i];
        cb2_[nt] = p.ffn_b2[e0 + 32 * nt + i];
    }

    // ---------------- attention of head c
    f32x16 s[NKT];
    zero_tiles<NKT>(s);
    for (int kc = 0; kc < (C >> 3); kc += 4) {   // S^T[key][query] = sum_ch K[key][ch] Q[query][ch]
        f32x4 qv[4], kv[4][NKT];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (FUSED) {
                qv[g] = ld4(qkv_t + (r0 + i) * ldq + c * C + 8 * (kc + g) + 4 * h2);
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {   // key tiles beyond the workgroup's rows alias its last tile (scores masked)
                    const int krow = 32 * (kt < nw ? kt : nw - 1) + i;
                    kv[g][kt] = ld4(qkv_t + krow * ldq + (p.fold ? 2 : 2 + c) * C + 8 * (kc + g) + 4 * h2);
                }
            } else {
                qv[g] = buf_ld4(r_qkv, q_off + 32u * (kc + g));
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) kv[g][kt] = buf_ld4(r_qkv, k_off[kt] + 32u * (kc + g));
            }
        }
        attn_scores_grp<NKT>(s, qv, kv);
    }
    struct VG { float v[4][NC]; };
    auto vfetch = [&](int f, VG& gq) __attribute__((always_inline)) {
        const int kt = f >> 2, r4 = (f & 3) << 2;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int key = 32 * kt + tile_row(r4 + rr, lane);
#pragma unroll
            for (int nt = 0; nt < NC; ++nt) {
                if constexpr (FUSED) gq.v[rr][nt] = qkv_t[(key < 32 * nw ? key : 32 * nw - 1) * ldq + (p.fold ? 2 : 4 + c) * C + 32 * nt + i];
                else gq.v[rr][nt] = buf_ld(r_qkv, v_base + (unsigned)((key * ld + 32 * nt) * 4));
            }
        }
    };
    VG v0, v1;
    vfetch(0, v0);
    vfetch(1, v1);
    ESMI_CT();   // S^T issued
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * kt + tile_row(r, lane);
            const float v = key < p.N ? s[kt][r] * p.scale : -INFINITY;
            s[kt][r] = v;
            mx = fmaxf(mx, v);
        }
    }
    mx = fmaxf(mx, swap32_f(mx));
    float den = 0.0f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = exp_fast_f32(s[kt][r] - mx);
            s[kt][r] = e;
            den += e;
        }
    }
    den += swap32_f(den);
    const float inv = rcp_fast_f32(den);
    f32x16 o[NC];
    zero_tiles<NC>(o);
#pragma unroll
    for (int f = 0; f < 4 * NKT; f += 2) {   // one 16-key step: accumulator registers [4f, 4f + 8) of key tile f >> 2
        f32x4 pa, pb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pa[e] = s[f >> 2][((f & 3) << 2) + e] * inv;
            pb[e] = s[f >> 2][((f & 3) << 2) + 4 + e] * inv;
        }
        VG w0 = v0, w1 = v1;
        if (f + 2 < 4 * NKT) vfetch(f + 2, v0);
        if (f + 3 < 4 * NKT) vfetch(f + 3, v1);
        attn_pv_step<NC>(o, pa, pb, w0.v, w1.v);
    }
    ESMI_CT();   // softmax + PV issued
    tile_store<NC>(buf, LD, c * C, o, lane);     // contexts of both heads side by side: proj's K dimension
    __syncthreads();
    // ---------------- proj (this wave's output columns, K = 2C), residual, LN1
    f32x16 y[NCH];
    zero_tiles<NCH>(y);
    wave_gemm_k<NCH, 2 * NC>(y, gp, a_row, true, p.proj_w, NC, 0, c * NCH, lane);
    WaveGrp<NEH> gm;
    wave_prefetch<NEH>(gm, p.ffn_w, NE, 0, c * NEH, lane);
    const unsigned mbits = (unsigned)ballot64(mb != 0);
    bool rz[16], rout[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = tile_row(r, lane);
        const int pos = t0 + row;
        rout[r] = pos < 0 || pos >= p.N;
        rz[r] = !rout[r] && ((mbits >> row) & 1u);
    }
#pragma unroll
    for (int nt = 0; nt < NCH; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) y[nt][r] += pb_[nt] + xres[nt][r];
    }
    layernorm_split<NCH>(y, g1_, be1_, stats, rt, c, lane);   // (its barriers also fence the proj reads of the tile)
#pragma unroll
    for (int nt = 0; nt < NCH; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (rz[r] || rout[r]) y[nt][r] = 0.0f;   // (rows outside the sequence: the MixFFN conv's zero padding; never stored)
    }
    tile_store<NCH>(buf, LD, c0, y, lane);
    __syncthreads();
    ESMI_CT();   // proj + LN1 + store
    // ---------------- MixFFN: (Linear folded into) dense conv k3 on the y1 rows of partner and neighbours -> GELU -> mlp2
    f32x16 m[NEH];
    ESMI_CT();   // (the stage of the separate Linear, gone)
    zero_tiles<NEH>(m);
    {
        const float* const taps[3] = {a_row - LD, a_row, a_row + LD};
        const bool tok[3] = {true, true, true};
        wave_gemm_taps<NEH, 3, NC, false>(m, gm, taps, tok, p.ffn_w, (long)EC * C, NE, 0, c * NEH, lane);
    }
    WaveGrp<NCH> g2;
    wave_prefetch<NCH>(g2, p.mlp2_w, NC, 0, c * NCH, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int pos = t0 + tile_row(r, lane);
#pragma unroll
        for (int nt = 0; nt < NEH; ++nt) {
            const float bias = cb_[nt] - (pos == 0 ? cb0_[nt] : 0.0f) - (pos == p.N - 1 ? cb2_[nt] : 0.0f);
            m[nt][r] = gelu_fast_f32(m[nt][r] + bias);
        }
    }
    __syncthreads();            // every wave has read what it needs of the hidden tile
    tile_store<NEH>(buf, LD, e0, m, lane);
    __syncthreads();
    ESMI_CT();   // conv + gelu + store
    f32x16 z[NCH];
    zero_tiles<NCH>(z);
    wave_gemm_k<NCH, NE>(z, g2, a_row, true, p.mlp2_w, NC, 0, c * NCH, lane);
#pragma unroll
    for (int nt = 0; nt < NCH; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) z[nt][r] += b2_[nt] + y[nt][r];
    }
    layernorm_split<NCH>(z, g2_, be2_, stats, rt, c, lane);
    ESMI_CT();   // LN2 done
    const int row_lo = p.halo, row_hi = 32 * nw - p.halo;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = tile_row(r, lane);
        const bool keep = r0 + row >= row_lo && r0 + row < row_hi && t0 + row >= 0;
        const unsigned off = keep ? (unsigned)(((t0 + row) * C + c0 + i) * 4) : kBufOOB;
#pragma unroll
        for (int nt = 0; nt < NCH; ++nt) buf_st(r_out, off + 128u * nt, rz[r] ? 0.0f : z[nt][r]);
    }
}

template <int NKT, int NC, int E, int NCI = 0, int KT = 1, int STRIDE = 1>
__global__ __launch_bounds__(128 * kEncSplitMaxTiles, kChainWps) void enc_attn_ffn_split_kernel(const EncAttnFfnP p) {
    enc_attn_ffn_split_body<NKT, NC, E, NCI, KT, STRIDE>(p);
}


}  // namespace esmi
