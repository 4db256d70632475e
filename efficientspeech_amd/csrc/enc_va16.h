// Fuse + variance adaptor (+ the decoder's phoneme-rate first stage, + the length regulator's scan) for dim = 32 models whose
// sequence one workgroup covers (T <= 128, two encoder levels): the round-5 form of enc_fuse_va.h -- same reference operations
// (layers/networks.py:189-219, :128-165, :346-384, :233-244 and the row-wise head of :291-294), built from chain16.h:
//
//   * ceil(T / 16) waves of one workgroup, a 16-row tile each (T = 128: 8 waves = two per SIMD), products transposed
//     (lane = row), v_mfma_f32_16x16x32_f16;
//   * the 172 KB of GEMM weights come into LDS ONCE per workgroup, a stage ahead, by LDS-DMA into two 36 KB halves:
//       Fuse (mlp 0, mlp 1, ConvTranspose taps, fuse Linear: 32 KB) -> A | conv1 of the three predictors (36 KB) -> B |
//       conv2 (36 KB) -> A | head K 0..63 (32 KB) -> B | head K 64..127 (32 KB) -> A
//     (enc_fuse_va_kernel streamed every matrix once per WAVE from L2: 4 x 172 KB per CU);
//   * the small parameter vectors are staged in LDS at entry (one round trip), the fused features and the predictors' hidden rows
//     live in LDS as f16 planes, the embedding rows enter the head straight from the tables;
//   * no instantiation spills (enc_fuse_va_kernel<1,3>: 256 VGPRs + 158 AGPRs of spill space).
//
// Row bookkeeping as in enc_fuse_va.h: rows >= T of the last tile are "outside" (conv zero padding: never stored), masked rows
// (padding phonemes) are computed and zeroed where the reference's masked_fill does.
#pragma once
#include "chain16.h"
#include "enc_fuse_va.h"

namespace esmi {

constexpr int kVa16MaxWaves = 8;
constexpr int kVa16Dim = 32;

// LDS map (floats / dwords)
struct Va16Lds {
    static constexpr int DIM = kVa16Dim;
    static constexpr int LDD = DIM + 4, LDT = 3 * DIM + 4, LDC = 2 * DIM + 4;
    static constexpr int ROWS = 16 * kVa16MaxWaves + 2;                 // shared tiles: one zero row in front and behind
    static constexpr int fused = 0;                                     // [ROWS][LDD] planes of the fused features
    static constexpr int region = fused + ROWS * LDD;                   // hidden [ROWS][LDT], aliased by the per-wave Fuse scratch
    static constexpr int priv = 16 * LDC + 17 * LDD;                    //   (cat [16][LDC] + tmp [17][LDD], row 16 of tmp = zeros)
    static constexpr int region_sz = (ROWS * LDT > kVa16MaxWaves * priv ? ROWS * LDT : kVa16MaxWaves * priv);   // and the duration-feature planes [16][LDD] per wave
    static constexpr int sdur = region + region_sz;                     // [128] ints
    static constexpr int par = sdur + 128;                              // parameter vectors, see PV_*: 32 slots of 32 floats
    static constexpr int par_sz = 32 * 32 + 4 * 128;                    // (the head's shift vector lands twice: its copy is a whole wave-instruction)
    static constexpr int emb = par + par_sz;                            // pitch / energy embedding tables [2][DIM][DIM]
    static constexpr int wA = emb + 2 * DIM * DIM;                      // weight halves
    static constexpr int half = 9 * 1024;                               // 36 KB
    static constexpr int wB = wA + half;
    static constexpr int total = wB + half;
};
static_assert(Va16Lds::total * 4 <= 160 * 1024, "enc_va16: LDS");
inline int va16_lds_bytes() { return Va16Lds::total * (int)sizeof(float); }

// parameter vectors in LDS (float offsets inside Va16Lds::par).  Staged by LDS-DMA, eight 32-float vectors per wave-instruction (lane l
// copies 16 bytes of vector l >> 3 -- the destination of an LDS-DMA is lane-linear, so eight consecutive slots are one instruction),
// the head's three 128-float vectors as two more instructions.
enum { PV_MLPB0 = 0, PV_MLPB1 = 32, PV_UPB1 = 64, PV_FUSEB = 96, PV_LN2G = 128, PV_LN2B = 160, PV_EDGE = 192 /* pitch, energy: 31 edges, +inf */,
       PV_PRED = 256 /* + 256 q: conv1_b, ln1_g, ln1_b, conv2_b, lin_w, 3 unused */,
       PV_HEADB = 1024, PV_HEADG = PV_HEADB + 128, PV_HEADBE = PV_HEADG + 128 };

template <int KU>   // ConvTranspose1d kernel (3: the Fuse weights must fit one LDS half)
__device__ __forceinline__ void enc_va16_body(const FuseVaP& p) {
    using namespace c16;
    typedef Va16Lds M;
    constexpr int DIM = M::DIM, LDD = M::LDD, LDT = M::LDT, LDC = M::LDC;
    static_assert((20 + 4 * KU) * 256 <= Va16Lds::half, "enc_va16: Fuse weights exceed an LDS half");
    ESMI_DYN_LDS(lds);
    ESMI_CT_INIT(2);
    ESMI_CT();   // entry
    const int nw = (int)(blockDim.x >> 6), w = uniform_i(wave_id());
    const int lane = lane_id(), i = lane & 15, g = lane >> 4;
    const int b = (int)blockIdx.x;
    const int r0 = 16 * w, pos = r0 + i;                    // this lane's row of the workgroup / position of the utterance
    unsigned* const ldu = reinterpret_cast<unsigned*>(lds);
    unsigned* const fusedT = ldu + M::fused;
    unsigned* const hidT = ldu + M::region;
    unsigned* const catP = ldu + M::region + w * M::priv;   // this wave's [16][LDC]
    unsigned* const tmpP = catP + 16 * LDC;                 //             [17][LDD]
    int* const sdur = reinterpret_cast<int*>(lds + M::sdur);
    float* const par = lds + M::par;
    float* const embT = lds + M::emb;
    float* const wA = lds + M::wA;
    float* const wB = lds + M::wB;
    const int wp = wpos(lane);
    const int lw1 = wlane(lane, 1), lw4 = wlane(lane, 4);
    const f32x4 z4 = zero4();
    const int rot = (int)blockIdx.x;          // request order of the weight fragments, see dma_frags

    // ---------------- entry: this lane's input rows requested first (the Fuse stage waits for them), then the first stage's weights and
    // every parameter vector on their way (LDS-DMA)
    const bool rout = pos >= p.T;
    const BufRsrc r_mask = make_rsrc(p.mask ? p.mask + (long)b * p.T : nullptr, p.T);
    const bool rz = !rout && buf_ld_u8(r_mask, (unsigned)pos) != 0;
    const BufRsrc r_feat = make_rsrc(p.feat ? p.feat + (long)b * p.T * 4 * DIM : nullptr, (long)p.T * 4 * DIM * 4);
    const unsigned frow = rout ? kBufOOB : (unsigned)(pos * 4 * DIM * 4);          // byte offset of this row of feat
    const BufRsrc r_pt = make_rsrc(p.pitch_t ? p.pitch_t + (long)b * p.T : nullptr, (long)p.T * 4);
    const BufRsrc r_et = make_rsrc(p.energy_t ? p.energy_t + (long)b * p.T : nullptr, (long)p.T * 4);
    const BufRsrc r_dt = make_rsrc(p.dur_t ? p.dur_t + (long)b * p.T : nullptr, (long)p.T * 4);
    const unsigned trow = rout ? kBufOOB : (unsigned)(pos * 4);
    const float tv_p = buf_ld(r_pt, trow), tv_e = buf_ld(r_et, trow), tv_d = buf_ld(r_dt, trow);   // teacher values (0 when absent)
    const float lb0 = p.pred[0].lin_b[0], lb1 = p.pred[1].lin_b[0], lb2 = p.pred[2].lin_b[0];
    const int e_i = (lane & 31) < DIM - 1 ? (lane & 31) : DIM - 2;
    const float edge_p = p.pred[0].bins[e_i], edge_e = p.pred[1].bins[e_i];      // bucket edges (dim - 1 of them), to LDS behind the entry barrier
    // Fuse, level 0 and level 1 operands straight from global memory
    const int n1 = p.n_i[1];
    const int n_base = floor_div(r0 - (KU - 1), 2);
    const int n_i = n_base + i;
    const BufRsrc r_f0 = make_rsrc(p.feats[0] + (long)b * p.n_i[0] * DIM, (long)p.n_i[0] * DIM * 4);
    const BufRsrc r_f1 = make_rsrc(p.feats[1] + (long)b * n1 * 2 * DIM, (long)n1 * 2 * DIM * 4);
    const unsigned o0 = rout ? kBufOOB : (unsigned)(pos * DIM * 4) + gl_lane(lane);
    const unsigned o1 = (n_i < 0 || n_i >= n1) ? kBufOOB : (unsigned)(n_i * 2 * DIM * 4) + gl_lane(lane);
    const f32x4 g0a = buf_ld4(r_f0, o0), g0b = buf_ld4(r_f0, o0 + 32u);                       // (split into the f16 pieces behind the entry barrier)
    const f32x4 g1a = buf_ld4(r_f1, o1), g1b = buf_ld4(r_f1, o1 + 32u), g1c = buf_ld4(r_f1, o1 + 128u), g1d = buf_ld4(r_f1, o1 + 160u);
    {
        // half A: mlp 0 (4 KiB) | mlp 1 (8) | ConvTranspose taps (4 KU) | fuse Linear (8)
        dma_frags(p.mlp_w[0], wA, 4, w, nw, lane, rot);
        dma_frags(p.mlp_w[1], wA + 4 * 256, 8, w, nw, lane, rot);
        dma_frags(p.up_w[1], wA + 12 * 256, 4 * KU, w, nw, lane, rot);
        dma_frags(p.fuse_w, wA + (12 + 4 * KU) * 256, 8, w, nw, lane, rot);
        ESMI_CT();   // (weights requested)
        // parameter vectors: instruction k copies slots 8k .. 8k + 7 (wave k % nw issues it)
        const int v8 = lane >> 3, c8 = 4 * (lane & 7);
        auto pick8 = [&](const float* a0, const float* a1, const float* a2, const float* a3, const float* a4, const float* a5, const float* a6,
                         const float* a7) __attribute__((always_inline)) {
            const float* lo = v8 & 1 ? (v8 & 2 ? a3 : a1) : (v8 & 2 ? a2 : a0);
            const float* hi = v8 & 1 ? (v8 & 2 ? a7 : a5) : (v8 & 2 ? a6 : a4);
            return (v8 & 4 ? hi : lo) + c8;
        };
        // (slots 6, 7 = the bucket edges: dim - 1 floats each, written from registers below -- a 32-float copy would read one float past
        // the arrays)
        if (w == 0 % nw) lds_dma16(pick8(p.mlp_b[0], p.mlp_b[1], p.up_b[1], p.fuse_b, p.pred[2].ln2_g, p.pred[2].ln2_b, p.fuse_b, p.fuse_b), par, lane);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const PredW& d = p.pred[q];
            if (w == (1 + q) % nw) lds_dma16(pick8(d.conv1_b, d.ln1_g, d.ln1_b, d.conv2_b, d.lin_w, d.lin_w, d.lin_w, d.lin_w), par + PV_PRED + 256 * q, lane);
        }
        if (p.h0) {   // head bias | gain (two 128-float vectors = one instruction), shift (half an instruction: lanes 32.. copy it again)
            if (w == 4 % nw) lds_dma16((lane < 32 ? p.head_b : p.head_g) + 4 * (lane & 31), par + PV_HEADB, lane);
            if (w == 5 % nw) lds_dma16(p.head_beta + 4 * (lane & 31), par + PV_HEADBE, lane);
        }
        if (w == 0) {                                       // zero rows around the fused tile, this wave's zero row for taps that
            if (lane < LDD) {                               // fall between input rows of the ConvTranspose
                fusedT[lane] = 0u;
                fusedT[(16 * nw + 1) * LDD + lane] = 0u;
            }
        }
        if (lane < LDD) tmpP[16 * LDD + lane] = 0u;
    }
    ESMI_CT();   // (rows requested)
    wait_vm0();
    ESMI_CT();   // (own queue drained)
    wg_sync_lds();              // half A, the parameter vectors and the zero rows are in place
    ESMI_CT();   // 1: entry loads landed
    const f16x2p a0 = split_f16x2(g0a, g0b), a10 = split_f16x2(g1a, g1b), a11 = split_f16x2(g1c, g1d);
    // the next stage's weights (conv1 of pitch | energy | duration: 12 KiB each) and the embedding tables start now, under the Fuse stage
#pragma unroll
    for (int q = 0; q < 3; ++q) dma_frags(p.pred[q].conv1_w, wB + q * 12 * 256, 12, w, nw, lane, rot);
    dma_frags(p.pred[0].emb, embT, 4, w, nw, lane, rot);
    dma_frags(p.pred[1].emb, embT + DIM * DIM, 4, w, nw, lane, rot);
    if (w == 0) {                                           // bucket edges, +inf behind the dim - 1 of them (read two barriers later)
        const int l5 = lane & 31;
        par[PV_EDGE + lane] = l5 < DIM - 1 ? (lane < 32 ? edge_p : edge_e) : INFINITY;
    }
    f32x4 acc[2];
    {   // level 0: Linear(dim, dim)
        acc[0] = z4; acc[1] = z4;
        gemm_pf<2, 1, 1>(acc, wA, lw1, 0, [&](int) __attribute__((always_inline)) { return a0; });
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f32x4 bc = ld4_lds(par + PV_MLPB0 + 16 * nt + 4 * g);
            planes_store(catP + i * LDC + wp, nt, LDC / 2 - 2, fmaf4(acc[nt], kF16WScaleInv, bc));
        }
    }
    {   // level 1: Linear(2 dim, dim) on rows n_base + i, then ConvTranspose1d(stride 2) cropped to T
        acc[0] = z4; acc[1] = z4;
        gemm_pf<2, 1, 2>(acc, wA + 4 * 256, lw1, 0, [&](int ks) __attribute__((always_inline)) { return ks ? a11 : a10; });
        const bool n_ok = n_i >= 0 && n_i < n1;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f32x4 bc = ld4_lds(par + PV_MLPB1 + 16 * nt + 4 * g);
            planes_store(tmpP + i * LDD + wp, nt, DIM / 2, n_ok ? fmaf4(acc[nt], kF16WScaleInv, bc) : z4);   // rows that do not exist contribute nothing
        }
        lds_wave_sync();
        acc[0] = z4; acc[1] = z4;
        WFrags<2> uf[KU];
        f16x2p ua[KU];
#pragma unroll
        for (int j = 0; j < KU; ++j) {      // out[2 n + j] += in[n] W_j
            const int q = pos - j, nq = q >> 1;
            const bool ok = q >= 0 && (q & 1) == 0 && nq < n1;
            ua[j] = planes_load(tmpP + (ok ? nq - n_base : 16) * LDD + 4 * g, 0, DIM / 2);
            wfrags_load<2, 1, 2>(uf[j], 0, wA + (12 + 4 * j) * 256, lw1, 0);
        }
        sched_fence();
#pragma unroll
        for (int j = 0; j < KU; ++j) mma_all<2>(acc, uf[j], ua[j]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f32x4 bc = ld4_lds(par + PV_UPB1 + 16 * nt + 4 * g);
            planes_store(catP + i * LDC + wp, 2 + nt, LDC / 2 - 2, fmaf4(acc[nt], kF16WScaleInv, bc));
        }
        lds_wave_sync();
    }
    f32x4 fz[2];                // the fused features of this lane's row (stored to feat at the end)
    {   // Linear(2 dim, dim) on the concatenation, masked_fill
        acc[0] = z4; acc[1] = z4;
        const unsigned* rowp = catP + i * LDC + 4 * g;
        gemm_pf<2, 1, 2>(acc, wA + (12 + 4 * KU) * 256, lw1, 0, [&](int ks) __attribute__((always_inline)) { return planes_load(rowp, ks, LDC / 2 - 2); });
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f32x4 bc = ld4_lds(par + PV_FUSEB + 16 * nt + 4 * g);
            fz[nt] = (rout || rz) ? z4 : fmaf4(acc[nt], kF16WScaleInv, bc);       // outside rows = the convs' zero padding
            planes_store(fusedT + (1 + pos) * LDD + wp, nt, DIM / 2, fz[nt]);
        }
    }
    ESMI_CT();   // (fuse computed)
    wait_vm0();                 // (this wave's share of conv1's weights and of the embedding tables)
    ESMI_CT();   // (queue drained)
    wg_sync_lds();              // fused rows of the neighbours in place; the Fuse scratch is dead in every wave; half A is free; half B has landed
    ESMI_CT();   // 2: fuse done
#pragma unroll
    for (int q = 0; q < 3; ++q) dma_frags(p.pred[q].conv2_w, wA + q * 12 * 256, 12, w, nw, lane, rot);
    if (w == 0) {               // zero rows around the hidden tile (they alias Fuse scratch)
        for (int c = lane; c < LDT; c += 64) {
            hidT[c] = 0u;
            hidT[(16 * nw + 1) * LDT + c] = 0u;
        }
    }
    // ---------------- conv1 (k = 3) of the three predictors on the shared fused rows, ReLU, LayerNorm, ReLU -> hidden planes
    f32x4 c[6];                 // [2 q + nt]
    {
#pragma unroll
        for (int n = 0; n < 6; ++n) c[n] = z4;
        const unsigned* rowp = fusedT + pos * LDD + 4 * g;                     // tile row of position pos - 1
        WFrags<6> wf[2];
        f16x2p av[2];
        auto fetch = [&](int j, int s) __attribute__((always_inline)) {
            av[s] = planes_load(rowp + j * LDD, 0, DIM / 2);
#pragma unroll
            for (int q = 0; q < 3; ++q) wfrags_load<2, 1, 6>(wf[s], 2 * q, wB + (q * 12 + 4 * j) * 256, lw1, 0);
        };
        fetch(0, 0);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j + 1 < 3) fetch(j + 1, (j + 1) & 1);
            sched_fence();
            mma_all<6>(c, wf[j & 1], av[j & 1]);
            sched_fence();
        }
        ESMI_CT();   // 3: conv1 issued
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            f32x4 gg[2], bb[2], v[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float* pv = par + PV_PRED + 256 * q + 16 * nt + 4 * g;
                const f32x4 b1 = ld4_lds(pv);
                gg[nt] = ld4_lds(pv + 32);
                bb[nt] = ld4_lds(pv + 64);
                v[nt] = relu4(fmaf4(c[2 * q + nt], kF16WScaleInv, b1));
            }
            layernorm<2>(v, gg, bb);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) planes_store(hidT + (1 + pos) * LDT + wp, 2 * q + nt, LDT / 2 - 2, rout ? z4 : relu4(v[nt]));
        }
    }
    ESMI_CT();   // (LN1 stored)
    wait_vm0();                 // (conv2's weights)
    ESMI_CT();   // (queue drained)
    wg_sync_lds();              // the neighbours' hidden rows (and the zero rows) in place; conv2's weights have landed; half B is free
    ESMI_CT();   // 4: LN1 + store
    if (p.h0) dma_frags(p.head_w, wB, 32, w, nw, lane, rot);     // head, K groups 0, 1
    // ---------------- conv2 (k = 3) on each predictor's own hidden rows, ReLU
    {
#pragma unroll
        for (int n = 0; n < 6; ++n) c[n] = z4;
        const unsigned* rowp = hidT + pos * LDT + 4 * g;
        WFrags<6> wf[2];
        f16x2p av[2][3];
        auto fetch = [&](int j, int s) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                av[s][q] = planes_load(rowp + j * LDT, q, LDT / 2 - 2);
                wfrags_load<2, 1, 6>(wf[s], 2 * q, wA + (q * 12 + 4 * j) * 256, lw1, 0);
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j + 1 < 3) fetch(j + 1, (j + 1) & 1);
            sched_fence();
            const WFrags<6>& f = wf[j & 1];
#pragma unroll
            for (int n = 0; n < 6; ++n) c[n] = mfma16_f16(f.w1[n], av[j & 1][n >> 1].h2, c[n]);
#pragma unroll
            for (int n = 0; n < 6; ++n) c[n] = mfma16_f16(f.w2[n], av[j & 1][n >> 1].h1, c[n]);
#pragma unroll
            for (int n = 0; n < 6; ++n) c[n] = mfma16_f16(f.w1[n], av[j & 1][n >> 1].h1, c[n]);
            sched_fence();
        }
    }
    ESMI_CT();   // (conv2 issued)
    wait_vm0();                 // (the head's first weight half)
    ESMI_CT();   // (queue drained)
    wg_sync_lds();              // every wave has read its neighbours' hidden rows: the region is free again; half A is free
    ESMI_CT();   // 5: conv2 done
    if (p.h0) dma_frags(p.head_w + 32 * 256, wA, 32, w, nw, lane, rot);   // head, K groups 2, 3
    // ---------------- Linear(dim, 1) on the pre-norm2 tensor, bucketize, embeddings, duration features
    float pr[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        float s = 0.0f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float* pv = par + PV_PRED + 256 * q + 16 * nt + 4 * g;
            c[2 * q + nt] = relu4(fmaf4(c[2 * q + nt], kF16WScaleInv, ld4_lds(pv + 96)));
            const f32x4 lwv = ld4_lds(pv + 128);
#pragma unroll
            for (int e = 0; e < 4; ++e) s = fmaf(c[2 * q + nt][e], lwv[e], s);
        }
        pr[q] = row_sum4(s) + (q == 0 ? lb0 : (q == 1 ? lb1 : lb2));
    }
    pr[2] = fmaxf(pr[2], 0.0f);
    int bidx[2];                // torch.bucketize(v, edges, right=False) = number of edges strictly below v
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const bool has_t = q == 0 ? p.pitch_t != nullptr : p.energy_t != nullptr;
        const float v = (has_t && !rout) ? (q == 0 ? tv_p : tv_e) : pr[q];
        const f32x4 e0 = ld4_lds(par + PV_EDGE + 32 * q + 8 * g), e1 = ld4_lds(par + PV_EDGE + 32 * q + 8 * g + 4);
        float cnt = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) cnt += (e0[e] < v ? 1.0f : 0.0f) + (e1[e] < v ? 1.0f : 0.0f);
        bidx[q] = (int)row_sum4(cnt);
    }
    // embedding rows from the LDS copies of the tables, in the head's B-operand order (k-step 1 = pitch, 2 = energy: the lane's eight
    // channels c0 .. c0 + 3, c0 + 8 .. c0 + 11) -- stored to feat from the same registers
    f32x4 em[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float* row = embT + q * DIM * DIM + bidx[q] * DIM + 16 * (g >> 1) + 4 * (g & 1);
        em[q][0] = rz ? z4 : ld4_lds(row);
        em[q][1] = rz ? z4 : ld4_lds(row + 8);
    }
    f32x4 df[2];                // duration features (networks.py:161-163)
    {
        f32x4 gg[2], bb[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            gg[nt] = ld4_lds(par + PV_LN2G + 16 * nt + 4 * g);
            bb[nt] = ld4_lds(par + PV_LN2B + 16 * nt + 4 * g);
            df[nt] = c[4 + nt];
        }
        layernorm<2>(df, gg, bb);
        if (rz) { df[0] = z4; df[1] = z4; }
    }
    float dval = p.dur_t ? (float)__builtin_bit_cast(int, tv_d) : rintf(pr[2]);   // torch.round: half to even
    if (p.mask) {                                                                  // networks.py:381-382
        if (rz) dval = 0.0f;
        dval = fmaxf(dval, 0.0f);
    }
    if (p.cum && g == 0) sdur[pos] = rout ? 0 : max((int)dval, 0);
    ESMI_CT();   // 6: predictions done
    // ---------------- outputs behind the last barrier that waits for the vector-memory queue (no barrier waits for a store)
    auto store_outputs = [&]() __attribute__((always_inline)) {
    if (p.feat) {               // (NULL: the lean inference call -- the decoder gathers h0, nobody reads the feature rows)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            buf_st4(r_feat, frow + (unsigned)((16 * nt + 4 * g) * 4), fz[nt]);
            buf_st4(r_feat, frow + (unsigned)((3 * DIM + 16 * nt + 4 * g) * 4), df[nt]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned off = frow + (unsigned)(((1 + q) * DIM + 16 * (g >> 1) + 4 * (g & 1)) * 4);
            buf_st4(r_feat, off, em[q][0]);
            buf_st4(r_feat, off + 32u, em[q][1]);
        }
    }
    {
        const unsigned srow = (!rout && g == 0) ? (unsigned)(pos * 4) : kBufOOB;   // one lane per row
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (p.preds[q]) {
                const BufRsrc r_pred = make_rsrc(p.preds[q] + (long)b * p.T, (long)p.T * 4);
                buf_st(r_pred, srow, pr[q]);
            }
        }
        if (p.pitch_idx) {
            const BufRsrc r_pi = make_rsrc(p.pitch_idx + (long)b * p.T, (long)p.T * 4);
            buf_st_i(r_pi, srow, bidx[0]);
        }
        if (p.energy_idx) {
            const BufRsrc r_ei = make_rsrc(p.energy_idx + (long)b * p.T, (long)p.T * 4);
            buf_st_i(r_ei, srow, bidx[1]);
        }
        const BufRsrc r_dur = make_rsrc(p.dur + (long)b * p.T, (long)p.T * 4);
        buf_st_i(r_dur, srow, (int)dval);
    }
    };
    f32x4 hh[8];
    if (p.h0) {
        // ---------------- decoder head at phoneme rate: h0 = LN(tanh(feat . Wp^T + b)), K = fused | pitch emb | energy emb | duration feats
        unsigned* const dfP = hidT + w * (16 * LDD);        // this wave's duration-feature planes [16][LDD] (the hidden tile is dead)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) planes_store(dfP + i * LDD + wp, nt, DIM / 2, df[nt]);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) hh[nt] = z4;
        const f16x2p e_p = split_f16x2(em[0][0], em[0][1]), e_e = split_f16x2(em[1][0], em[1][1]);
        WFrags<8> hf[2];
        // (the first weight half was requested behind the conv1 barrier and landed at the conv2 barrier; the second is waited for below)
        wfrags_load<8, 4, 8>(hf[0], 0, wB, lw4, 0);
        const f16x2p a_f = planes_load(fusedT + (1 + pos) * LDD + 4 * g, 0, DIM / 2);
        wfrags_load<8, 4, 8>(hf[1], 0, wB, lw4, 1);
        sched_fence();
        mma_all<8>(hh, hf[0], a_f);
        sched_fence();
        mma_all<8>(hh, hf[1], e_p);
        ESMI_CT();   // (head first half issued)
        wait_vm0();
        ESMI_CT();   // (queue drained)
        wg_sync_lds();          // second half landed (every wave's share); this wave's duration-feature planes are written
        store_outputs();
        // (half A holds K groups 2, 3 as its groups 0, 1: the matrix's group stride is 4 NTW slots either way)
        wfrags_load<8, 4, 8>(hf[0], 0, wA, lw4, 0);
        const f16x2p a_d = planes_load(dfP + i * LDD + 4 * g, 0, DIM / 2);
        wfrags_load<8, 4, 8>(hf[1], 0, wA, lw4, 1);
        sched_fence();
        mma_all<8>(hh, hf[0], e_e);
        sched_fence();
        mma_all<8>(hh, hf[1], a_d);
        ESMI_CT();   // 7: head GEMM
        f32x4 gg[8], bb[8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            const f32x4 hb = ld4_lds(par + PV_HEADB + 16 * nt + 4 * g);
            gg[nt] = ld4_lds(par + PV_HEADG + 16 * nt + 4 * g);
            bb[nt] = ld4_lds(par + PV_HEADBE + 16 * nt + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) hh[nt][e] = tanh_fast_f32(fmaf(hh[nt][e], kF16WScaleInv, hb[e]));
        }
        layernorm<8>(hh, gg, bb);
        ESMI_CT();   // 8: head done
    }
    if (!p.h0) store_outputs();
    if (p.h0) {
        const BufRsrc r_h0 = make_rsrc(p.h0 + (long)b * p.T * 4 * DIM, (long)p.T * 4 * DIM * 4);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) buf_st4(r_h0, frow + (unsigned)((16 * nt + 4 * g) * 4), hh[nt]);
    }
    ESMI_CT();   // 9: outputs issued
    if (p.cum) {   // FeatureUpsampler's scan (networks.py:233-244) while the durations are still on the CU; T <= 128 here
        wg_sync_lds();
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

template <int KU>
__global__ __launch_bounds__(64 * kVa16MaxWaves, 2) void enc_va16_kernel(const FuseVaP p) {
    enc_va16_body<KU>(p);
}

}  // namespace esmi
