// Whole encoder blocks of dim = 32 models as round-5 chain kernels (chain16.h: 16-row tiles, two waves per SIMD, weights once per
// workgroup through LDS, transposed products).  Same reference operations as enc_attn_ffn.h (layers/networks.py:54-85,
// layers/blocks.py:22-29,43-71) with the pack-time folds of round 4 (embedding tables per conv tap, q_h = x M_h with keys = values = x,
// MixFFN's Linear inside its k = 3 conv):
//
//   enc_b0_16_kernel<NKT>: block 0 (C = 32, one head, composed k = 3 merge conv as three table gathers), N <= 16 NKT <= 128 positions:
//                          one wave per 16 positions.
//   enc_b1_16_kernel:      block 1 (C = 64, two heads, k = 1 stride-2 merge conv from 32 channels), N <= 64 positions: two waves per
//                          16 positions -- wave c computes head c's attention, then half of the output columns of every later GEMM;
//                          the LayerNorms merge per-row (mean, M2) pairs of the two halves (as enc_attn_ffn_split_kernel does).
//
// Attention in this layout: S^T = K Q^T with the keys as the first MFMA operand, so lane (j, g) holds query j's scores of keys
// 16 kt + 4 g + (0..3) -- softmax is in-lane plus the two lane-group exchanges -- and the probabilities of two key tiles ARE the lane's
// eight k-slots of a 32-key step of P V (no shuffle).  V^T comes from LDS planes `xT[channel][key]` whose key order is permuted to that
// slot order (written once per wave through a 16 x 32 fp32 transposing tile).
#pragma once
#include "chain16.h"
#include "enc_attn_ffn.h"

namespace esmi {

// ------------------------------------------------------------------------------------------------ block 0
struct B016Lds {
    static constexpr int C = 32, LD = C + 4, NK = 128, LDX = NK + 4;
    static constexpr int xP = 0;                          // [128][LD] planes of x (keys / the q GEMM's rows)
    static constexpr int xT = xP + 128 * LD;              // [C][LDX] planes of x^T, key order permuted (see above)
    static constexpr int y1P = xT + C * LDX;              // [130][LD] planes of y1, zero rows around (MixFFN conv)
    static constexpr int priv = y1P + 130 * LD;           // per wave: fp32 transposing tile [16][LD] + planes [16][LD] (q, then ctx, then hidden)
    static constexpr int priv_sz = 2 * 16 * LD;
    static constexpr int par = priv + 8 * priv_sz;        // 16 slots of 32 floats
    static constexpr int wts = par + 512;                 // M | O | MixFFN conv taps | mlp2: 24 KiB
    static constexpr int total = wts + 24 * 256;
};
static_assert(B016Lds::total * 4 <= 160 * 1024, "enc_b0_16: LDS");
enum { B0_PROJB = 0, B0_LN1G = 32, B0_LN1B = 64, B0_FFNB = 96, B0_FFNB0 = 128, B0_FFNB2 = 160, B0_MLP2B = 192, B0_LN2G = 224, B0_LN2B = 256 };

template <int NKT>   // key tiles of 16: N <= 16 NKT
__device__ __forceinline__ void enc_b0_16_body(const EncAttnFfnP& p) {
    using namespace c16;
    typedef B016Lds M;
    constexpr int C = M::C, LD = M::LD, LDX = M::LDX;
    ESMI_DYN_LDS(lds);
    ESMI_CT_INIT(0);
    ESMI_CT();   // entry
    const int nw = (int)(blockDim.x >> 6), w = uniform_i(wave_id());
    const int lane = lane_id(), i = lane & 15, g = lane >> 4;
    const int b = (int)blockIdx.x, rot = b;
    const int pos = 16 * w + i;
    unsigned* const ldu = reinterpret_cast<unsigned*>(lds);
    unsigned* const xP = ldu + M::xP;
    unsigned* const xT = ldu + M::xT;
    unsigned* const y1P = ldu + M::y1P;
    float* const xf = lds + M::priv + w * M::priv_sz;                 // fp32 [16][LD]
    unsigned* const qP = ldu + M::priv + w * M::priv_sz + 16 * LD;    // planes [16][LD]
    float* const par = lds + M::par;
    float* const wM = lds + M::wts, * const wO = wM + 4 * 256, * const wF = wO + 4 * 256, * const w2 = wF + 12 * 256;
    const int wp = wpos(lane), lw1 = wlane(lane, 1);
    const f32x4 z4 = zero4();
    // ---------------- entry: ids first (the table rows depend on them), weights and parameter vectors by LDS-DMA
    const int* idb = p.m.ids + (long)b * p.m.n_in;
    int id[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int ti = pos + j - 1;
        const bool ok = ti >= 0 && ti < p.m.n_in;
        const int v = idb[ok ? ti : 0];
        id[j] = !ok ? -1 : ((v < 0 || v >= p.m.vocab) ? 0 : v);       // (the reference raises IndexError; stay in bounds)
    }
    dma_frags(p.m.qkv_w, wM, 4, w, nw, lane, rot);
    dma_frags(p.proj_w, wO, 4, w, nw, lane, rot);
    dma_frags(p.ffn_w, wF, 12, w, nw, lane, rot);
    dma_frags(p.mlp2_w, w2, 4, w, nw, lane, rot);
    {
        const int v8 = lane >> 3, c8 = 4 * (lane & 7);
        auto pick8 = [&](const float* a0, const float* a1, const float* a2, const float* a3, const float* a4, const float* a5, const float* a6,
                         const float* a7) __attribute__((always_inline)) {
            const float* lo = v8 & 1 ? (v8 & 2 ? a3 : a1) : (v8 & 2 ? a2 : a0);
            const float* hi = v8 & 1 ? (v8 & 2 ? a7 : a5) : (v8 & 2 ? a6 : a4);
            return (v8 & 4 ? hi : lo) + c8;
        };
        if (w == 0 % nw) lds_dma16(pick8(p.proj_b, p.ln1_g, p.ln1_b, p.ffn_b, p.ffn_b0, p.ffn_b2, p.mlp2_b, p.ln2_g), par, lane);
        if (w == 1 % nw) lds_dma16(p.ln2_b + c8, par + B0_LN2B, lane);   // (slot 8; the other lanes' copies of it fill the unused slots 9..15)
    }
    if (w == 0 && lane < LD) {                                        // zero rows around the y1 tile
        y1P[lane] = 0u;
        y1P[(16 * nw + 1) * LD + lane] = 0u;
    }
    if (nw < NKT) {   // short sequence: the key tiles beyond the workgroup's rows read zeros (their scores are masked, P = 0)
        for (int e = (int)threadIdx.x; e < C * LDX; e += (int)blockDim.x) xT[e] = 0u;
        for (int e = (int)threadIdx.x + 16 * nw * LD; e < 16 * NKT * LD; e += (int)blockDim.x) xP[e] = 0u;
        wg_sync_lds();
    }
    const bool rout = pos >= p.N;
    unsigned mb = 0;                // blocks.py:51-57: the mask is padded with True and max-pooled by the block's total stride
    {
        const BufRsrc r_mask = make_rsrc(p.mask ? p.mask + (long)b * p.mask_len : nullptr, p.mask_len);
        for (int q = 0; q < p.mask_pool; ++q) {
            const int idx = pos * p.mask_pool + q;
            mb |= buf_ld_u8(r_mask, (unsigned)idx) | (unsigned)(p.mask && idx >= p.mask_len);
        }
    }
    const bool rz = !rout && mb != 0;
    ESMI_CT();   // 1: requests issued
    // ---------------- x[t] = sum_j E_j[id[t + j - 1]]: three table rows per position, straight into the D^T layout
    f32x4 xacc[2] = {z4, z4};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float* row = p.m.emb_conv + ((long)j * p.m.vocab + (id[j] < 0 ? 0 : id[j])) * C + 4 * g;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f32x4 v = ld4(row + 16 * nt);
            if (id[j] >= 0) xacc[nt] += v;
        }
    }
    if (rout) { xacc[0] = z4; xacc[1] = z4; }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        planes_store(xP + pos * LD + wp, nt, C / 2, xacc[nt]);
        *reinterpret_cast<f32x4*>(xf + i * LD + 16 * nt + 4 * g) = xacc[nt];
    }
    lds_wave_sync();
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {   // lane (channel i of tile ct, g): keys 4 g + (0..3) of this wave's tile -> x^T planes
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = xf[(4 * g + r) * LD + 16 * ct + i];
        unsigned h1a, h2a, h1b, h2b;
        split_f16_pair(v[0], v[1], h1a, h2a);
        split_f16_pair(v[2], v[3], h1b, h2b);
        unsigned* dst = xT + (16 * ct + i) * LDX + 16 * (w >> 1) + 4 * g + 2 * (w & 1);
        *reinterpret_cast<u32x2*>(dst) = u32x2{h1a, h1b};
        *reinterpret_cast<u32x2*>(dst + M::NK / 2) = u32x2{h2a, h2b};
    }
    ESMI_CT();   // 2: x gathered
    wait_vm0();
    wg_sync_lds();              // x planes / x^T of every wave, the weights and the parameter vectors are in place
    ESMI_CT();   // 3: barrier
    // ---------------- q = x M  (scores = q x^T: the folded q / k projection)
    {
        f32x4 q[2] = {z4, z4};
        const unsigned* rowp = xP + pos * LD + 4 * g;
        gemm_pf<2, 1, 1>(q, wM, lw1, 0, [&](int) __attribute__((always_inline)) { return planes_load(rowp, 0, C / 2); });
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) planes_store(qP + i * LD + wp, nt, C / 2, q[nt] * kF16WScaleInv);
        lds_wave_sync();
    }
    // ---------------- S^T[key][query] = K Q^T, softmax over the keys of this lane's query
    f32x4 s[NKT];
    {
        const f16x2p qf = planes_load(qP + i * LD + 4 * g, 0, C / 2);
        f16x2p kf[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) kf[kt] = planes_load(xP + (16 * kt + i) * LD + 4 * g, 0, C / 2);
        sched_fence();
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) s[kt] = mfma16_f16(kf[kt].h2, qf.h1, z4);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) s[kt] = mfma16_f16(kf[kt].h1, qf.h2, s[kt]);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) s[kt] = mfma16_f16(kf[kt].h1, qf.h1, s[kt]);
    }
    ESMI_CT();   // 4: scores
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * g + r;
            const float v = key < p.N ? s[kt][r] * p.scale : -INFINITY;
            s[kt][r] = v;
            mx = fmaxf(mx, v);
        }
    }
    mx = row_max4(mx);
    float den = 0.0f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = exp_fast_f32(s[kt][r] - mx);
            s[kt][r] = e;
            den += e;
        }
    }
    const float inv = rcp_fast_f32(row_sum4(den));
    ESMI_CT();   // 5: softmax
    // ---------------- ctx^T[channel][query] = V^T P^T, 32 keys per step
    f32x4 o[2] = {z4, z4};
    {
        f16x2p vf[NKT / 2][2];
#pragma unroll
        for (int ks = 0; ks < NKT / 2; ++ks) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const unsigned* vp = xT + (16 * ct + i) * LDX + 16 * ks + 4 * g;
                vf[ks][ct].h1 = *reinterpret_cast<const u32x4*>(vp);
                vf[ks][ct].h2 = *reinterpret_cast<const u32x4*>(vp + M::NK / 2);
            }
        }
        sched_fence();
#pragma unroll
        for (int ks = 0; ks < NKT / 2; ++ks) {
            const f16x2p pf = split_f16x2(s[2 * ks] * inv, s[2 * ks + 1] * inv);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) o[ct] = mfma16_f16(vf[ks][ct].h2, pf.h1, o[ct]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) o[ct] = mfma16_f16(vf[ks][ct].h1, pf.h2, o[ct]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) o[ct] = mfma16_f16(vf[ks][ct].h1, pf.h1, o[ct]);
        }
    }
    ESMI_CT();   // 6: P V
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) planes_store(qP + i * LD + wp, nt, C / 2, o[nt]);
    lds_wave_sync();
    // ---------------- y1 = mask(LN1(ctx O + bias + x))
    f32x4 y[2] = {z4, z4};
    {
        const unsigned* rowp = qP + i * LD + 4 * g;
        gemm_pf<2, 1, 1>(y, wO, lw1, 0, [&](int) __attribute__((always_inline)) { return planes_load(rowp, 0, C / 2); });
        f32x4 gg[2], bb[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f32x4 pb = ld4_lds(par + B0_PROJB + 16 * nt + 4 * g);
            gg[nt] = ld4_lds(par + B0_LN1G + 16 * nt + 4 * g);
            bb[nt] = ld4_lds(par + B0_LN1B + 16 * nt + 4 * g);
            y[nt] = fmaf4(y[nt], kF16WScaleInv, pb) + xacc[nt];
        }
        layernorm<2>(y, gg, bb);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (rz || rout) y[nt] = z4;      // (rows outside the sequence: the MixFFN conv's zero padding; never stored)
            planes_store(y1P + (1 + pos) * LD + wp, nt, C / 2, y[nt]);
        }
    }
    ESMI_CT();   // 7: proj + LN1
    wg_sync_lds();              // the neighbouring waves' boundary rows (and the zero rows) are in place
    ESMI_CT();   // 8: barrier
    // ---------------- MixFFN: (Linear folded into) dense conv k3 -> GELU -> mlp2, residual, LN2, mask
    {
        f32x4 m[2] = {z4, z4};
        WFrags<2> ff[3];
        f16x2p fa[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            fa[j] = planes_load(y1P + (pos + j) * LD + 4 * g, 0, C / 2);
            wfrags_load<2, 1, 2>(ff[j], 0, wF + 4 * j * 256, lw1, 0);
        }
        sched_fence();
#pragma unroll
        for (int j = 0; j < 3; ++j) mma_all<2>(m, ff[j], fa[j]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f32x4 cb = ld4_lds(par + B0_FFNB + 16 * nt + 4 * g), cb0 = ld4_lds(par + B0_FFNB0 + 16 * nt + 4 * g),
                        cb2 = ld4_lds(par + B0_FFNB2 + 16 * nt + 4 * g);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float bias = cb[e] - (pos == 0 ? cb0[e] : 0.0f) - (pos == p.N - 1 ? cb2[e] : 0.0f);
                v[e] = gelu_fast_f32(fmaf(m[nt][e], kF16WScaleInv, bias));
            }
            planes_store(qP + i * LD + wp, nt, C / 2, v);
        }
        lds_wave_sync();
        f32x4 z[2] = {z4, z4};
        const unsigned* rowp = qP + i * LD + 4 * g;
        gemm_pf<2, 1, 1>(z, w2, lw1, 0, [&](int) __attribute__((always_inline)) { return planes_load(rowp, 0, C / 2); });
        f32x4 gg[2], bb[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f32x4 b2 = ld4_lds(par + B0_MLP2B + 16 * nt + 4 * g);
            gg[nt] = ld4_lds(par + B0_LN2G + 16 * nt + 4 * g);
            bb[nt] = ld4_lds(par + B0_LN2B + 16 * nt + 4 * g);
            z[nt] = fmaf4(z[nt], kF16WScaleInv, b2) + y[nt];
        }
        layernorm<2>(z, gg, bb);
        const BufRsrc r_out = make_rsrc(p.out + (long)b * p.N * C, (long)p.N * C * 4);
        const unsigned off = rout ? kBufOOB : (unsigned)(pos * C * 4);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) buf_st4(r_out, off + (unsigned)((16 * nt + 4 * g) * 4), rz ? z4 : z[nt]);
    }
    ESMI_CT();   // 9: done
}

template <int NKT>
__global__ __launch_bounds__(64 * 8, 2) void enc_b0_16_kernel(const EncAttnFfnP p) {
    enc_b0_16_body<NKT>(p);
}

// ------------------------------------------------------------------------------------------------ block 1
struct B116Lds {
    static constexpr int C = 64, LD = C + 4, CIN = 32, NK = 64, LDX = NK + 4, LDC = 2 * C + 4;
    static constexpr int r1 = 0;                          // x planes [64][LD] (keys / q rows), later y1 planes [66][LD] (zero rows around)
    static constexpr int r2 = r1 + 66 * LD;               // x^T planes [C][LDX], later the MixFFN hidden planes [64][LD]
    static constexpr int r3 = r2 + 64 * LD;               // per wave [16][LD] (fp32 transposing tile, then q planes), later ctx planes [64][LDC] of both heads
    static constexpr int r3_sz = 8 * 16 * LD;
    static constexpr int stats = r3 + r3_sz;              // [4 row tiles][2 waves][16 rows][2]
    static constexpr int par = stats + 256;               // 12 slots of 64 floats
    static constexpr int wY = par + 768;                  // merge conv + q matrix (40 KiB), later the MixFFN conv taps (48 KiB)
    static constexpr int wX = wY + 48 * 256;              // attention output matrix (32 KiB), later mlp2 (16 KiB)
    static constexpr int total = wX + 32 * 256;
};
static_assert(64 * B116Lds::LDC <= B116Lds::r3_sz && B116Lds::total * 4 <= 160 * 1024, "enc_b1_16: LDS");
enum { B1_PROJB = 0, B1_LN1G = 64, B1_LN1B = 128, B1_FFNB = 192, B1_FFNB0 = 256, B1_FFNB2 = 320, B1_MLP2B = 384, B1_LN2G = 448, B1_LN2B = 512 };

// LayerNorm over C = 64 channels of which this wave holds 32 (D^T layout, two 16-channel tiles); the partner wave of the row tile holds
// the other 32: per-row (mean, M2) pairs cross through `st` ([2][16][2] floats of this row tile) and merge with the parallel-variance
// formula, operands in the same order in both waves (bit-identical statistics).  One workgroup barrier inside; the caller guarantees
// that the previous use of `st` was consumed (a barrier since).
__device__ __forceinline__ void layernorm_pair16(f32x4 (&v)[2], const f32x4 (&gg)[2], const f32x4 (&bb)[2], float* st, int c, int i, int g,
                                                 float eps = 1e-5f) {
    using namespace c16;
    float s = (v[0][0] + v[0][1]) + (v[0][2] + v[0][3]) + ((v[1][0] + v[1][1]) + (v[1][2] + v[1][3]));
    const float mean_l = row_sum4(s) * (1.0f / 32.0f);
    float q = 0.0f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[nt][e] - mean_l;
            q = fmaf(d, d, q);
        }
    }
    const float m2_l = row_sum4(q);
    if (g == 0) *reinterpret_cast<f32x2*>(st + (c * 16 + i) * 2) = f32x2{mean_l, m2_l};
    wg_sync_lds();
    const f32x2 o = *reinterpret_cast<const f32x2*>(st + ((c ^ 1) * 16 + i) * 2);
    const float ma = c == 0 ? mean_l : o[0], mb = c == 0 ? o[0] : mean_l;
    const float qa = c == 0 ? m2_l : o[1], qb = c == 0 ? o[1] : m2_l;
    const float mean = 0.5f * (ma + mb);
    const float d = mb - ma;
    const float m2 = (qa + qb) + d * d * 16.0f;                      // n_a n_b / (n_a + n_b) = 32 * 32 / 64
    const float rstd = rsqrt_fast_f32(m2 * (1.0f / 64.0f) + eps);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[nt][e] = fmaf((v[nt][e] - mean) * rstd, gg[nt][e], bb[nt][e]);
    }
}

__device__ __forceinline__ void enc_b1_16_body(const EncAttnFfnP& p) {
    using namespace c16;
    typedef B116Lds M;
    constexpr int C = M::C, LD = M::LD, LDX = M::LDX, LDC = M::LDC, CIN = M::CIN, NKT = 4;
    ESMI_DYN_LDS(lds);
    ESMI_CT_INIT(1);
    ESMI_CT();   // entry
    const int nw = (int)(blockDim.x >> 6), w = uniform_i(wave_id());
    const int nrt = nw >> 1, rt = w >> 1, c = w & 1;                  // row tiles, this wave's tile and head / column half
    const int lane = lane_id(), i = lane & 15, g = lane >> 4;
    const int b = (int)blockIdx.x, rot = b;
    const int pos = 16 * rt + i;
    unsigned* const ldu = reinterpret_cast<unsigned*>(lds);
    unsigned* const xP = ldu + M::r1;
    unsigned* const y1P = ldu + M::r1;
    unsigned* const xT = ldu + M::r2;
    unsigned* const hidP = ldu + M::r2;
    float* const xf = lds + M::r3 + w * (16 * LD);                    // fp32 [16][36] (this wave's 32 channels of x)
    unsigned* const qP = ldu + M::r3 + w * (16 * LD);                 // planes [16][LD]: q of head c
    unsigned* const ctxP = ldu + M::r3;                               // planes [64][LDC]: contexts of both heads side by side
    float* const st = lds + M::stats + rt * 64;
    float* const par = lds + M::par;
    float* const wY = lds + M::wY, * const wX = lds + M::wX;
    const int wp = wpos(lane), lw2 = wlane(lane, 2), lw4 = wlane(lane, 4);
    const f32x4 z4 = zero4();
    // ---------------- entry
    const bool rout = pos >= p.N;
    const BufRsrc r_in = make_rsrc(p.m.x_in + (long)b * p.m.n_in * CIN, (long)p.m.n_in * CIN * 4);
    const f16x2p a_in = global_bop(r_in, rout ? kBufOOB : (unsigned)(2 * pos * CIN * 4) + gl_lane(lane), 0);   // stride 2, k = 1, no padding
    dma_frags(p.m.merge_w, wY, 8, w, nw, lane, rot);
    dma_frags(p.m.qkv_w, wY + 8 * 256, 32, w, nw, lane, rot);
    dma_frags(p.proj_w, wX, 32, w, nw, lane, rot);
    {
        const int v4 = lane >> 4, c4 = 4 * (lane & 15);
        auto pick4 = [&](const float* a0, const float* a1, const float* a2, const float* a3) __attribute__((always_inline)) {
            return (v4 & 2 ? (v4 & 1 ? a3 : a2) : (v4 & 1 ? a1 : a0)) + c4;
        };
        if (w == 0 % nw) lds_dma16(pick4(p.proj_b, p.ln1_g, p.ln1_b, p.ffn_b), par, lane);
        if (w == 1 % nw) lds_dma16(pick4(p.ffn_b0, p.ffn_b2, p.mlp2_b, p.ln2_g), par + 256, lane);
        if (w == 2 % nw) lds_dma16(p.ln2_b + c4, par + 512, lane);
    }
    if (nrt < NKT) {   // short sequence: the key tiles beyond the workgroup's rows read zeros (their scores are masked, P = 0)
        for (int e = (int)threadIdx.x; e < C * LDX; e += (int)blockDim.x) xT[e] = 0u;
        for (int e = (int)threadIdx.x + 16 * nrt * LD; e < 16 * NKT * LD; e += (int)blockDim.x) xP[e] = 0u;
        wg_sync_lds();
    }
    unsigned mb = 0;                // blocks.py:51-57: the mask is padded with True and max-pooled by the block's total stride
    {
        const BufRsrc r_mask = make_rsrc(p.mask ? p.mask + (long)b * p.mask_len : nullptr, p.mask_len);
        for (int q = 0; q < p.mask_pool; ++q) {
            const int idx = pos * p.mask_pool + q;
            mb |= buf_ld_u8(r_mask, (unsigned)idx) | (unsigned)(p.mask && idx >= p.mask_len);
        }
    }
    const bool rz = !rout && mb != 0;
    wait_vm0();
    wg_sync_lds();              // weights (merge conv, q matrix, output matrix) and parameter vectors have landed
    ESMI_CT();   // 1: entry loads landed
    // ---------------- x = composed merge conv (k = 1, stride 2): this wave's 32 channels of its row tile
    f32x4 xh[2] = {z4, z4};
    gemm_pf<2, 2, 1>(xh, wY + c * 256, lw2, 0, [&](int) __attribute__((always_inline)) { return a_in; });
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        xh[nt] = rout ? z4 : xh[nt] * kF16WScaleInv;
        planes_store(xP + pos * LD + wp, 2 * c + nt, C / 2, xh[nt]);
        *reinterpret_cast<f32x4*>(xf + i * (CIN + 4) + 16 * nt + 4 * g) = xh[nt];
    }
    lds_wave_sync();
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {   // lane (channel i of this wave's tile ct, g): keys 4 g + (0..3) of row tile rt -> x^T planes
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = xf[(4 * g + r) * (CIN + 4) + 16 * ct + i];
        unsigned h1a, h2a, h1b, h2b;
        split_f16_pair(v[0], v[1], h1a, h2a);
        split_f16_pair(v[2], v[3], h1b, h2b);
        unsigned* dst = xT + (32 * c + 16 * ct + i) * LDX + 16 * (rt >> 1) + 4 * g + 2 * (rt & 1);
        *reinterpret_cast<u32x2*>(dst) = u32x2{h1a, h1b};
        *reinterpret_cast<u32x2*>(dst + M::NK / 2) = u32x2{h2a, h2b};
    }
    wg_sync_lds();              // x planes / x^T of every wave are in place
    ESMI_CT();   // 2: merge conv
    // ---------------- q of head c = x M_c (64 columns, K = 64)
    {
        f32x4 q[4] = {z4, z4, z4, z4};
        const unsigned* rowp = xP + pos * LD + 4 * g;
        gemm_pf<4, 4, 2>(q, wY + 8 * 256 + 2 * c * 256, lw4, 0, [&](int ks) __attribute__((always_inline)) { return planes_load(rowp, ks, C / 2); });
        lds_wave_sync();        // (the transposing tile shares the q planes' space: its reads are done)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) planes_store(qP + i * LD + wp, nt, C / 2, q[nt] * kF16WScaleInv);
        lds_wave_sync();
    }
    // ---------------- head c: S^T[key][query] = K Q^T over 64 channels, softmax, ctx^T = V^T P^T
    f32x4 s[NKT];
    {
        f16x2p qf[2], kf[2][NKT];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = planes_load(qP + i * LD + 4 * g, ks, C / 2);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) kf[ks][kt] = planes_load(xP + (16 * kt + i) * LD + 4 * g, ks, C / 2);
        }
        sched_fence();
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) s[kt] = z4;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) s[kt] = mfma16_f16(kf[ks][kt].h2, qf[ks].h1, s[kt]);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) s[kt] = mfma16_f16(kf[ks][kt].h1, qf[ks].h2, s[kt]);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) s[kt] = mfma16_f16(kf[ks][kt].h1, qf[ks].h1, s[kt]);
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * g + r;
            const float v = key < p.N ? s[kt][r] * p.scale : -INFINITY;
            s[kt][r] = v;
            mx = fmaxf(mx, v);
        }
    }
    mx = row_max4(mx);
    float den = 0.0f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = exp_fast_f32(s[kt][r] - mx);
            s[kt][r] = e;
            den += e;
        }
    }
    const float inv = rcp_fast_f32(row_sum4(den));
    f32x4 o[4] = {z4, z4, z4, z4};
    {
        f16x2p vf[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const unsigned* vp = xT + (16 * ct + i) * LDX + 16 * ks + 4 * g;
                vf[ks][ct].h1 = *reinterpret_cast<const u32x4*>(vp);
                vf[ks][ct].h2 = *reinterpret_cast<const u32x4*>(vp + M::NK / 2);
            }
        }
        sched_fence();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f16x2p pf = split_f16x2(s[2 * ks] * inv, s[2 * ks + 1] * inv);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) o[ct] = mfma16_f16(vf[ks][ct].h2, pf.h1, o[ct]);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) o[ct] = mfma16_f16(vf[ks][ct].h1, pf.h2, o[ct]);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) o[ct] = mfma16_f16(vf[ks][ct].h1, pf.h1, o[ct]);
        }
    }
    ESMI_CT();   // 3: attention
    wg_sync_lds();              // every wave is through with q, the x planes and x^T, and with the q matrix (weight half Y)
#pragma unroll
    for (int j = 0; j < 3; ++j) dma_frags(p.ffn_w + (long)j * C * C, wY + j * 16 * 256, 16, w, nw, lane, rot);   // MixFFN conv taps -> Y
    if (w == 0) {               // zero rows around the y1 tile (it takes the x planes' place)
        for (int e = lane; e < LD; e += 64) {
            y1P[e] = 0u;
            y1P[(16 * nrt + 1) * LD + e] = 0u;
        }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) planes_store(ctxP + pos * LDC + wp, 4 * c + ct, C, o[ct]);   // contexts of both heads side by side: the projection's K
    wg_sync_lds();
    ESMI_CT();   // 4: contexts exchanged
    // ---------------- proj (this wave's 32 output columns, K = 2 C), residual, LN1, mask
    f32x4 y[2] = {z4, z4};
    {
        const unsigned* rowp = ctxP + pos * LDC + 4 * g;
        gemm_pf<2, 2, 4>(y, wX + c * 256, lw2, 0, [&](int ks) __attribute__((always_inline)) { return planes_load(rowp, ks, C); });
        f32x4 gg[2], bb[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = 32 * c + 16 * nt + 4 * g;
            gg[nt] = ld4_lds(par + B1_LN1G + col);
            bb[nt] = ld4_lds(par + B1_LN1B + col);
            y[nt] = fmaf4(y[nt], kF16WScaleInv, ld4_lds(par + B1_PROJB + col)) + xh[nt];
        }
        layernorm_pair16(y, gg, bb, st, c, i, g);      // (its barrier: every wave is through with the ctx planes and the output matrix)
    }
    dma_frags(p.mlp2_w, wX, 16, w, nw, lane, rot);      // mlp2 -> X
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        if (rz || rout) y[nt] = z4;                     // (rows outside the sequence: the MixFFN conv's zero padding; never stored)
        planes_store(y1P + (1 + pos) * LD + wp, 2 * c + nt, C / 2, y[nt]);
    }
    wait_vm0();                 // (the MixFFN conv taps)
    wg_sync_lds();              // y1 rows of partner and neighbours (and the zero rows) are in place; Y holds the conv taps
    ESMI_CT();   // 5: proj + LN1
    // ---------------- MixFFN: (Linear folded into) dense conv k3 -> GELU -> mlp2, residual, LN2, mask
    {
        f32x4 m[2] = {z4, z4};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned* rowp = y1P + (pos + j) * LD + 4 * g;
            gemm_pf<2, 2, 2>(m, wY + j * 16 * 256 + c * 256, lw2, 0, [&](int ks) __attribute__((always_inline)) { return planes_load(rowp, ks, C / 2); });
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = 32 * c + 16 * nt + 4 * g;
            const f32x4 cb = ld4_lds(par + B1_FFNB + col), cb0 = ld4_lds(par + B1_FFNB0 + col), cb2 = ld4_lds(par + B1_FFNB2 + col);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float bias = cb[e] - (pos == 0 ? cb0[e] : 0.0f) - (pos == p.N - 1 ? cb2[e] : 0.0f);
                v[e] = gelu_fast_f32(fmaf(m[nt][e], kF16WScaleInv, bias));
            }
            planes_store(hidP + pos * LD + wp, 2 * c + nt, C / 2, v);
        }
    }
    wait_vm0();                 // (mlp2)
    wg_sync_lds();              // hidden rows of the partner are in place
    ESMI_CT();   // 6: conv + gelu
    {
        f32x4 z[2] = {z4, z4};
        const unsigned* rowp = hidP + pos * LD + 4 * g;
        gemm_pf<2, 2, 2>(z, wX + c * 256, lw2, 0, [&](int ks) __attribute__((always_inline)) { return planes_load(rowp, ks, C / 2); });
        f32x4 gg[2], bb[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = 32 * c + 16 * nt + 4 * g;
            gg[nt] = ld4_lds(par + B1_LN2G + col);
            bb[nt] = ld4_lds(par + B1_LN2B + col);
            z[nt] = fmaf4(z[nt], kF16WScaleInv, ld4_lds(par + B1_MLP2B + col)) + y[nt];
        }
        layernorm_pair16(z, gg, bb, st, c, i, g);
        const BufRsrc r_out = make_rsrc(p.out + (long)b * p.N * C, (long)p.N * C * 4);
        const unsigned off = rout ? kBufOOB : (unsigned)(pos * C * 4);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) buf_st4(r_out, off + (unsigned)((32 * c + 16 * nt + 4 * g) * 4), rz ? z4 : z[nt]);
    }
    ESMI_CT();   // 7: done
}

__global__ __launch_bounds__(64 * 8, 2) void enc_b1_16_kernel(const EncAttnFfnP p) {
    enc_b1_16_body(p);
}

}  // namespace esmi
