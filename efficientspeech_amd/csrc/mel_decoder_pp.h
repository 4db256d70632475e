// Mel decoder, dx2 = 128, role-alternating ("ping-pong") form -- MelDecoder.forward, layers/networks.py:291-304, with the length
// regulator gather (networks.py:233-244) in front and Phoneme2Mel's final masked_fill (networks.py:424-427) behind, for the
// case the tiny model runs in: the first stage h0 = LN(tanh(proj(x))) supplied at PHONEME rate (enc_fuse_va_kernel).
//
// Same arithmetic as mel_decoder_fold.h (LayerNorm folded into its consumer, statistics from the accumulators, skip tensor in the
// accumulator layout; read that file's header first), different orchestration.  The window form runs every wave through
// depthwise -> K loop -> tanh -> store, all waves in the same phase between workgroup barriers, so the matrix pipe idles while the
// VALU / LDS phases run and vice versa (rounds 1-3: matrix pipe 26 % busy, time ~ the SUM of the pipes' times).  Here ONE persistent
// 8-wave workgroup per CU (256 VGPRs per wave, 159 KB of LDS) holds TWO windows, one per 4-wave "slot"; SIMD s hosts wave s of slot 0
// and wave s of slot 1.  A step is four barrier-separated quarters; in every step one slot is in the M role and the other in the D role:
//   M (matrix) role, layer l: the wave owns all 128 rows x 32 columns; its whole weight slice (64 VGPRs) is register-resident, the row
//     tiles are the outer loop, and bias + tanh + statistics of row tile j-1 are interleaved with the MFMAs of row tile j
//     (quarter q = K loop of tile q).  The results stay in the accumulators across the step boundary.
//   D (data) role, the step after: quarter 0 finishes the M step (block end: LN + skip add on the accumulators; rows -> tile),
//     quarter 1 merges the rows' statistics and fetches the halo rows, quarters 2-3 stream the normalise + depthwise conv + operand
//     split of its 32 rows (all 128 channels), in place.  Meanwhile the other slot's waves run their K loops: MFMA from one wave,
//     VALU / LDS from the other on every SIMD.
// A window is 2 n_layers + 2 steps: [gather + conv 0] M0 [conv 1] M1 ... [normalise for mel] M_mel, alternating D / M; slot 1 runs one
// step behind slot 0.  The workgroup walks a static list of windows (XCD-aware: workgroup id % 8 is the XCD, an utterance's windows
// go to workgroups of one XCD), so there is no dispatch tail and the pipeline fills once per launch.
#pragma once
#include "mel_decoder_fold.h"

namespace esmi {

constexpr int kPpThreads = 512;
template <int KD>
__host__ __device__ constexpr int dec_pp_lds_floats() {
    // two tiles | group A x 2 | group B x 2 | partial statistics x 2 | block-end statistics (shared)
    return 2 * (kDecRows + 2 * kDecPadRows) * (128 + 4) + 2 * (KD + 1) * 128 + 2 * 5 * 128 + 2 * kDecRows * 8 + kDecRows * 8;
}

struct PpWin {   // one window of one utterance (wave-uniform)
    int b, f0, f_lo, out_hi, mlen, valid_end;
    bool edge;   // some rows lie outside [0, L)
};

template <int KD>
__global__ __launch_bounds__(kPpThreads, 1) void mel_decoder_pp_kernel(const MelDecP p) {
    constexpr int DX2 = 128, LDSROW = DX2 + 4, PAD = KD / 2;
    constexpr int TILE_F = (kDecRows + 2 * kDecPadRows) * LDSROW;
    constexpr int GA = (KD + 1) * DX2, GB = 5 * DX2, P_C0 = KD * DX2;
    constexpr int P_PWB = 0, P_G = DX2, P_B = 2 * DX2, P_SG = 3 * DX2, P_SB = 4 * DX2;
    constexpr int RS = 16, NR = RS + 2 * PAD;       // D role: a half wave = a strip of 16 rows x 128 channels
    constexpr float WSI = kF16WScaleInv;
    ESMI_DYN_LDS(lds);
    const int tid = (int)threadIdx.x, lane = lane_id();
#ifdef ESMI_WAVESIM
    const int w = wave_id();
#else
    const int w = __builtin_amdgcn_readfirstlane(wave_id());    // wave-uniform: slot / column slice / LDS bases stay in scalar registers
#endif
    const int slot = w >> 2, ns = w & 3;
    const int stid = tid & 255;                     // thread index inside the slot
    // (lane-derived indices are re-derived where they are used -- `LANE_IH` -- so that nothing but the lane id lives across the
    // quarters: at 3 x 64 resident registers every long-lived address register was a spill, and a spill reload costs ~1k cycles here)
#define LANE_IH const int ln_ = opaque_i(lane); const int i = ln_ & 31, h = ln_ >> 5; (void)i; (void)h
    float* xs = lds + slot * TILE_F;                                   // this slot's tile [132][LDSROW]
    float* pa = lds + 2 * TILE_F + slot * GA;                          // group A: folded taps, c0
    float* pbuf = lds + 2 * TILE_F + 2 * GA + slot * GB;               // group B: pw_b, ln_g, ln_b, skip_g, skip_b
    float* pst = lds + 2 * TILE_F + 2 * GA + 2 * GB + slot * (kDecRows * 8);   // [128][4][2] partial statistics of the M step's rows
    float* pu = lds + 2 * TILE_F + 2 * GA + 2 * GB + 2 * (kDecRows * 8);       // the same for u = LN(t) + skip (block ends); shared: the slots' D steps alternate

    const int n_layers = p.n_blocks * p.block_depth;
    const int n_stage = 2 * n_layers + 2;
    const int L = p.lmax_dev ? *p.lmax_dev : (p.lmax_host >= 0 ? p.lmax_host : batch_max_len(p.mel_len, p.B));
    const BufRsrc brs = make_rsrc(p.blob, p.lay.total * (long)sizeof(float));
    auto blob_ld = [&](long float_off, unsigned voff) __attribute__((always_inline)) { return buf_ld4s(brs, voff, (unsigned)(float_off * 4)); };

    // ---- the workgroup's window list: XCD x = id % 8 serves utterances b = 8u + x; its windows q = u * n_tiles + tile are dealt
    // round-robin to the XCD's workgroups; the workgroup's k-th window goes to slot k & 1
    const int xcd = (int)blockIdx.x & 7, jx = (int)blockIdx.x >> 3, wg_per_xcd = (int)gridDim.x >> 3;
    const int nq = ((p.B - xcd + 7) >> 3) * p.n_tiles;
    const int n_list = nq > jx ? (nq - jx + wg_per_xcd - 1) / wg_per_xcd : 0;   // windows of this workgroup
    auto window_at = [&](int k, PpWin& wn) __attribute__((always_inline)) {       // k-th window of the workgroup's list
        const int q = jx + k * wg_per_xcd;
        const int u = q / p.n_tiles, tile = q - u * p.n_tiles;
        wn.b = 8 * u + xcd;
        wn.mlen = p.mel_len ? min(p.mel_len[wn.b], L) : L;
        wn.f_lo = tile * p.TL;
        wn.f0 = wn.f_lo - p.halo;
        wn.out_hi = min(wn.f_lo + p.TL, p.L_out);
        wn.valid_end = p.apply_mask ? wn.mlen : L;
        wn.edge = wn.f0 < 0 || wn.f0 + kDecRows > L;
    };

    // ---- parameter staging through a register, by the slot's 256 threads: `fetch` issues the global load where the slots' last
    // reader has passed a barrier, `commit` writes LDS at the end of the same quarter (the load's latency is covered by the quarter's
    // work; an LDS-DMA here made hipcc wait for the transfer in front of the next LDS read, which it cannot tell apart from the target)
    auto fetch = [&](long float_off, int n4) __attribute__((always_inline)) {
        const int t4 = opaque_i(stid);
        return t4 < n4 ? blob_ld(float_off, (unsigned)t4 * 16u) : zero4();
    };
    auto commit = [&](float* dst, int n4, const f32x4& v) __attribute__((always_inline)) {
        const int t4 = opaque_i(stid);
        if (t4 < n4) reinterpret_cast<f32x4*>(dst)[t4] = v;
    };
    auto off_A = [&](int l) __attribute__((always_inline)) { return l == 0 ? p.lay.layer0_h0 : p.lay.layer0 + (long)l * p.lay.layer_stride; };
    auto off_B = [&](int l) __attribute__((always_inline)) {   // "layer" n_layers: the folded mel bias -> the pw_b slots
        return l < n_layers ? p.lay.layer0 + (long)l * p.lay.layer_stride + p.lay.l_pwb : p.lay.mel_b;
    };

    // zero pad rows of both tiles (never written again)
    for (int e = tid; e < 2 * 2 * kDecPadRows * LDSROW; e += kPpThreads) {
        const int t = e / (2 * kDecPadRows * LDSROW), r0 = e - t * (2 * kDecPadRows * LDSROW);
        const int r = r0 / LDSROW, c = r0 - r * LDSROW;
        const int rr = r < kDecPadRows ? r : kDecRows + r;
        lds[t * TILE_F + rr * LDSROW + c] = 0.0f;
    }

    // ================================================================== registers that live across steps
    f32x16 acc[4];          // M role: row tile j; lane (i, h) holds frame 32 j + i, channels 32 ns + 8 g + 4 h + e in [4 g + e]
    f32x16 skip[4];         // the skip tensor, same layout
    u32x4 W[8][2];          // the wave's weight slice: 16-channel step s, plane pl
    auto load_W = [&](long off) __attribute__((always_inline)) {   // off: float offset of the packed matrix
        const long wsl = off + (long)ns * 8 * 2 * 256;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) W[s][pl] = __builtin_bit_cast(u32x4, blob_ld(wsl + (s * 2 + pl) * 256, (unsigned)opaque_i(lane) * 16u));
        }
    };

    // ================================================================== M role pieces
    // bias + tanh + one-pass statistics of row tile J, slice S (accumulator elements 2S, 2S + 1); the per-lane pivot keeps the
    // one-pass sums of squares free of cancellation
    float st_c = 0.0f, st_s1 = 0.0f, st_s2 = 0.0f;
    f32x4 bq = zero4();
    auto epi_slice = [&](auto jc, auto sc) __attribute__((always_inline)) {
        constexpr int J = decltype(jc)::value, S = decltype(sc)::value;
        LANE_IH;
        if constexpr ((S & 1) == 0)
            bq = *reinterpret_cast<const f32x4*>(pbuf + P_PWB + opaque_i(32 * ns + 4 * h) + 8 * (S >> 1)) * kTanhExpScale;
#pragma unroll
        for (int e = 2 * S; e < 2 * S + 2; ++e) {
            const float t = tanh_fast_fma_f32(acc[J][e], WSI * kTanhExpScale, bq[e & 3]);
            acc[J][e] = t;
            if (e == 0) { st_c = t; st_s1 = 0.0f; st_s2 = 0.0f; }
            else {
                const float d = t - st_c;
                st_s1 += d;
                st_s2 = fmaf(d, d, st_s2);
            }
        }
    };
    // statistics of one row's 32 channels from the two half waves' one-pass sums (16 values each) -> dst[row][ns] = (mean, M2)
    auto stats_finish = [&](int j, float* dst) __attribute__((always_inline)) {
        LANE_IH;
        const float mh_ = fmaf(st_s1, 1.0f / 16.0f, st_c);                 // half mean
        const float m2h = fmaf(-st_s1 * (1.0f / 16.0f), st_s1, st_s2);     // half M2
        const float mo = swap32_f(mh_), m2o = swap32_f(m2h);
        const float dm = mh_ - mo;
        const float mean = 0.5f * (mh_ + mo), m2 = fmaf(8.0f * dm, dm, m2h + m2o);
        if (h == 0) *reinterpret_cast<f32x2*>(dst + (32 * j + i) * 8 + 2 * ns) = f32x2{mean, m2};
    };
    // K loop of row tile J over all 128 channels (8 steps x 3 MFMAs), the A fragments one step ahead; EPI: interleave the
    // epilogue of tile J - 1
    auto k_tile = [&](auto jc, auto epic) __attribute__((always_inline)) {
        constexpr int J = decltype(jc)::value;
        constexpr bool EPI = decltype(epic)::value;
        LANE_IH;
        const unsigned* a_base = reinterpret_cast<const unsigned*>(xs) + opaque_i((kDecPadRows + 32 * J + i) * LDSROW + 4 * h);
        // Two accumulator chains (even / odd steps): anything issued between two MFMAs on the SAME accumulator costs ~43 cycles
        // (MI355X_MICROARCH.md), and every step has A-fragment reads and an epilogue slice to place.
        f16x2p a[2];
        a[0].h1 = *reinterpret_cast<const u32x4*>(a_base);
        a[0].h2 = *reinterpret_cast<const u32x4*>(a_base + DX2 / 2);
        f32x16 odd = zero16();
        acc[J] = zero16();
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s + 1 < 8) {
                a[(s + 1) & 1].h1 = *reinterpret_cast<const u32x4*>(a_base + 8 * (s + 1));
                a[(s + 1) & 1].h2 = *reinterpret_cast<const u32x4*>(a_base + 8 * (s + 1) + DX2 / 2);
            }
            if (s & 1) odd = mfma32_split2_wx(W[s][0], W[s][1], a[s & 1], odd);
            else acc[J] = mfma32_split2_wx(W[s][0], W[s][1], a[s & 1], acc[J]);
            if constexpr (EPI) {
                if (s == 0) epi_slice(std::integral_constant<int, J - 1>{}, std::integral_constant<int, 0>{});
                if (s == 1) epi_slice(std::integral_constant<int, J - 1>{}, std::integral_constant<int, 1>{});
                if (s == 2) epi_slice(std::integral_constant<int, J - 1>{}, std::integral_constant<int, 2>{});
                if (s == 3) epi_slice(std::integral_constant<int, J - 1>{}, std::integral_constant<int, 3>{});
                if (s == 4) epi_slice(std::integral_constant<int, J - 1>{}, std::integral_constant<int, 4>{});
                if (s == 5) epi_slice(std::integral_constant<int, J - 1>{}, std::integral_constant<int, 5>{});
                if (s == 6) epi_slice(std::integral_constant<int, J - 1>{}, std::integral_constant<int, 6>{});
                if (s == 7) epi_slice(std::integral_constant<int, J - 1>{}, std::integral_constant<int, 7>{});
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[J][e] += odd[e];
        if constexpr (EPI) stats_finish(J - 1, pst);
    };
    auto epi_tile3 = [&]() __attribute__((always_inline)) {
        epi_slice(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
        epi_slice(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
        epi_slice(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});
        epi_slice(std::integral_constant<int, 3>{}, std::integral_constant<int, 3>{});
        epi_slice(std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{});
        epi_slice(std::integral_constant<int, 3>{}, std::integral_constant<int, 5>{});
        epi_slice(std::integral_constant<int, 3>{}, std::integral_constant<int, 6>{});
        epi_slice(std::integral_constant<int, 3>{}, std::integral_constant<int, 7>{});
        stats_finish(3, pst);
    };

    // ================================================================== D role pieces (accumulator layout)
    auto merge_row = [&](const float* prow, float& r, float& m) __attribute__((always_inline)) {   // Chan merge of a row's 4 partials
        const f32x4 a = *reinterpret_cast<const f32x4*>(prow), c = *reinterpret_cast<const f32x4*>(prow + 4);
        const float mu = ((a[0] + a[2]) + (c[0] + c[2])) * 0.25f;
        const float d0 = a[0] - mu, d1 = a[2] - mu, d2 = c[0] - mu, d3 = c[2] - mu;
        float dd = d0 * d0;
        dd = fmaf(d1, d1, dd);
        dd = fmaf(d2, d2, dd);
        dd = fmaf(d3, d3, dd);
        const float m2 = fmaf(32.0f, dd, (a[1] + a[3]) + (c[1] + c[3]));
        r = rsqrt_fast_f32(fmaf(m2, 1.0f / DX2, 1e-5f));
        m = -mu * r;
    };
    auto merge_own = [&](const float* src, float (&r)[4], float (&m)[4]) __attribute__((always_inline)) {
        LANE_IH;
#pragma unroll
        for (int j = 0; j < 4; ++j) merge_row(src + opaque_i(i * 8) + 32 * j * 8, r[j], m[j]);
    };
    // dst = LN(acc) [+ skip]: gain / shift at gp / bp, row statistics (r, m)
    auto ln_acc = [&](f32x16 (&dst)[4], const float* gp, const float* bp, const float (&r)[4], const float (&m)[4], auto add_c) __attribute__((always_inline)) {
        constexpr bool ADD = decltype(add_c)::value;
        LANE_IH;
        const int c0 = opaque_i(32 * ns + 4 * h);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 gg = *reinterpret_cast<const f32x4*>(gp + c0 + 8 * g), bb = *reinterpret_cast<const f32x4*>(bp + c0 + 8 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y = fmaf(fmaf(acc[j][4 * g + e], r[j], m[j]), gg[e], bb[e]);
                    dst[j][4 * g + e] = ADD ? y + skip[j][4 * g + e] : y;
                }
            }
        }
    };
    // two-pass partial statistics of the accumulators' rows over this wave's 32 channels -> dst[row][ns]
    auto stats_acc = [&](float* dst) __attribute__((always_inline)) {
        LANE_IH;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) s += (acc[j][4 * g] + acc[j][4 * g + 1]) + (acc[j][4 * g + 2] + acc[j][4 * g + 3]);
            s += swap32_f(s);
            const float mean = s * (1.0f / 32.0f);
            float q = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float d = acc[j][e] - mean;
                q = fmaf(d, d, q);
            }
            q += swap32_f(q);
            if (h == 0) *reinterpret_cast<f32x2*>(dst + (32 * j + i) * 8 + 2 * ns) = f32x2{mean, q};
        }
    };
    // accumulators -> tile (raw rows); rows outside [0, L) (edge windows) as -b/g of the consumer's LayerNorm (mel_decoder_fold.h)
    auto store_acc = [&](const PpWin& wn, const float* gp, const float* bp) __attribute__((always_inline)) {
        LANE_IH;
        const int c0 = opaque_i(32 * ns + 4 * h);
        float* base = xs + opaque_i((kDecPadRows + i) * LDSROW + 32 * ns + 4 * h);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 q = zero4();
            if (wn.edge) {
                const f32x4 gg = *reinterpret_cast<const f32x4*>(gp + c0 + 8 * g), bb = *reinterpret_cast<const f32x4*>(bp + c0 + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = -bb[e] * rcp_fast_f32(gg[e]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[j][4 * g + e];
                if (wn.edge) {
                    const int f = wn.f0 + 32 * j + i;
                    if (f < 0 || f >= L) v = q;
                }
                *reinterpret_cast<f32x4*>(base + 32 * j * LDSROW + 8 * g) = v;
            }
        }
    };
    // mel rows of the window `wn` from the accumulators (folded bias in the pw_b slots), masked store
    auto mel_store = [&](const PpWin& wn) __attribute__((always_inline)) {
        LANE_IH;
        if (32 * ns >= p.n_mel) return;
        const bool vec_ok = (p.n_mel & 3) == 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = wn.f0 + 32 * j + i;
            if (f < wn.f_lo || f >= wn.out_hi) continue;
            float* orow = p.mel + ((long)wn.b * p.L_out + f) * p.n_mel;
            const bool live = f < wn.valid_end;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = 32 * ns + 8 * g + 4 * h;
                if (col >= p.n_mel) continue;
                const f32x4 bc = *reinterpret_cast<const f32x4*>(pbuf + P_PWB + col);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = live ? fmaf(acc[j][4 * g + e], WSI, bc[e]) : 0.0f;
                if (vec_ok) {
                    *reinterpret_cast<f32x4*>(orow + col) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < p.n_mel) orow[col + e] = v[e];
                }
            }
        }
    };
    // first stage of a window: h0 rows of this wave's 32 frames -> tile; skip = the window's h0 rows x this wave's columns.
    // Source row per frame: phoneme row (>= 0), padding frame (-2: LN(tanh(proj_b)), packed), outside the sequence (-1: zeros).
    auto gather = [&](const PpWin& wn) __attribute__((always_inline)) {
        LANE_IH;
        int srow[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int f = wn.f0 + lane + 64 * k;
            int s;
            if (f < 0 || f >= L) s = -1;
            else if (f < wn.mlen) {
                const int ph = frame_to_phoneme(p.cum + wn.b * p.T, p.T, f);
                s = ph < p.T ? wn.b * p.T + ph : -2;
            } else s = -2;
            srow[k] = s;
        }
        // branch-free: a row that is not a phoneme row reads out of range (0 from the buffer hardware); padding frames then take the
        // packed LN(tanh(proj_b)) vector.  All loads of a pass are in flight together.
        const BufRsrc hrs = make_rsrc(p.h0, (long)p.B * p.T * DX2 * 4);
        auto row_off = [&](int s, int col) __attribute__((always_inline)) { return s >= 0 ? (unsigned)(s * (DX2 * 4) + 4 * col) : kBufOOB; };
        {
            const f32x4 padv = blob_ld(p.lay.h0_pad, (unsigned)(16 * i));
            f32x4 v[16];
            int sv[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int r = 32 * ns + 2 * it + h;
                sv[it] = shfl_i(ns < 2 ? srow[0] : srow[1], r & 63);
                v[it] = buf_ld4(hrs, row_off(sv[it], 4 * i));
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int r = 32 * ns + 2 * it + h;
                *reinterpret_cast<f32x4*>(xs + (kDecPadRows + r) * LDSROW + 4 * i) = sv[it] == -2 ? padv : v[it];
            }
        }
        {
            f32x4 padq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) padq[g] = blob_ld(p.lay.h0_pad, (unsigned)(4 * (32 * ns + 8 * g + 4 * h)));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int s = shfl_i(j < 2 ? srow[0] : srow[1], (32 * j + i) & 63);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = buf_ld4(hrs, row_off(s, 32 * ns + 8 * g + 4 * h));
#pragma unroll
                    for (int e = 0; e < 4; ++e) skip[j][4 * g + e] = s == -2 ? padq[g][e] : v[e];
                }
            }
        }
    };

    // ---- the consumer side (mel_decoder_fold.h `consume`), here per wave: rows [32 ns, +32) of the slot's tile, a half wave per
    // 16-row strip, split in three parts around the step's barriers
    struct Strip { f32x4 win[NR]; f32x4 tap[KD]; f32x4 tb; };
    auto strip_r0 = [&](int h) __attribute__((always_inline)) { return 32 * ns + 16 * h; };
    // part 1 (ahead of the halo barrier): statistics of the strip's rows -> pad floats; halo rows -> registers
    auto cons_head = [&](const PpWin& wn, auto ident_c, bool unit, const float* stsrc, Strip& S) __attribute__((always_inline)) {
        constexpr bool IDENT = decltype(ident_c)::value;
        constexpr int HP = IDENT ? 0 : PAD, NRW = RS + 2 * HP;
        LANE_IH;
        const int dw_cg = i, r0 = strip_r0(h);
        if (dw_cg < NRW) {
            const int pr = r0 - HP + dw_cg, f = wn.f0 + pr;
            float r = 0.0f, m = 0.0f;
            if (pr >= 0 && pr < kDecRows) {
                if (unit || f < 0 || f >= L) r = 1.0f;
                else merge_row(stsrc + pr * 8, r, m);
            }
            *reinterpret_cast<f32x2*>(xs + (kDecPadRows + pr) * LDSROW + DX2) = f32x2{r, m};
        }
        if constexpr (!IDENT) {
            const float* col = xs + opaque_i((kDecPadRows + r0 - HP) * LDSROW + 4 * dw_cg);
#pragma unroll
            for (int q = 0; q < HP; ++q) {
                S.win[q] = *reinterpret_cast<const f32x4*>(col + q * LDSROW);
                S.win[NRW - 1 - q] = *reinterpret_cast<const f32x4*>(col + (NRW - 1 - q) * LDSROW);
            }
        } else {
            lds_wave_sync();
        }
    };
    // rows [R0, R1) of the strip.  R0 == 0 first brings the strip's 16 raw rows and the window's statistics into registers and
    // normalises (the D role has the registers: the accumulators are dead), so that everything behind is register arithmetic with
    // the rows' chains independent of each other -- there is no second wave on the SIMD's VALU to hide a row-serial chain's latency.
    // The planes are written in place: all reads of the strip's rows precede the first write in program order (one half wave per strip).
    auto cons_rows = [&](auto ident_c, auto r0c, auto r1c, Strip& S) __attribute__((always_inline)) {
        constexpr bool IDENT = decltype(ident_c)::value;
        constexpr int R0 = decltype(r0c)::value, R1 = decltype(r1c)::value;
        constexpr int HP = IDENT ? 0 : PAD, NRW = RS + 2 * HP;
        LANE_IH;
        const int dw_cg = i, r0 = strip_r0(h);
        if constexpr (R0 == 0) {
            const float* col = xs + opaque_i((kDecPadRows + r0 - HP) * LDSROW + 4 * dw_cg);
            const float* stp = xs + opaque_i((kDecPadRows + r0 - HP) * LDSROW + DX2);
#pragma unroll
            for (int q = HP; q < HP + RS; ++q) S.win[q] = *reinterpret_cast<const f32x4*>(col + q * LDSROW);
            if constexpr (!IDENT) {
                const float* pat = pa + opaque_i(4 * dw_cg);
#pragma unroll
                for (int j = 0; j < KD; ++j) S.tap[j] = *reinterpret_cast<const f32x4*>(pat + j * DX2);
                S.tb = *reinterpret_cast<const f32x4*>(pat + P_C0);
            }
#pragma unroll
            for (int q = 0; q < NRW; ++q) {
                const f32x2 st = *reinterpret_cast<const f32x2*>(stp + q * LDSROW);
#pragma unroll
                for (int e = 0; e < 4; ++e) S.win[q][e] = fmaf(S.win[q][e], st[0], st[1]);
            }
            wave_lockstep();
        }
        unsigned* prow = reinterpret_cast<unsigned*>(xs) + opaque_i((kDecPadRows + r0) * LDSROW + 2 * dw_cg);
#pragma unroll
        for (int r = R0; r < R1; ++r) {
            f32x4 a;
            if constexpr (IDENT) a = S.win[r];
            else {
                a = S.tb;
#pragma unroll
                for (int j = 0; j < KD; ++j) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = fmaf(S.win[r + j][e], S.tap[j][e], a[e]);
                }
            }
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            unsigned h1a, h2a, h1b, h2b;
            split_f16_pair(a[0], a[1], h1a, h2a);
            split_f16_pair(a[2], a[3], h1b, h2b);
            unsigned* rowp = prow + r * LDSROW;
            *reinterpret_cast<u32x2*>(rowp) = u32x2{h1a, h1b};
            *reinterpret_cast<u32x2*>(rowp + DX2 / 2) = u32x2{h2a, h2b};
        }
    };
    typedef std::true_type TrueC;
    typedef std::false_type FalseC;
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    typedef std::integral_constant<int, 8> I8;
    typedef std::integral_constant<int, 16> I16;

    // ================================================================== the schedule
    // The workgroup's list is taken 64 windows at a time.  Lane k classifies window k: all padding (the final masked_fill or the
    // [L, L_out) tail zeroes it: filled here, never scheduled) or live.  The live windows go alternately to slot 0 and slot 1; a slot
    // runs, per window, the fixed sequence  D(0) M(0) D(1) M(1) ... D(n) M(n)  (D(0) = gather + conv 0; stage n = the mel Linear), slot 1
    // one step behind slot 0, then one D-type step that stores the last window's mel rows.  Every step is four barriers for all waves.
#ifdef ESMI_DEC_TRACE      // development: shader-clock stamps behind every barrier of workgroup 0, [wave][512]
    int tr_n = 0;
    const bool tr_on = p.trace && blockIdx.x == 8 && lane == 0;
#define PP_SYNC() do { if (tr_on && tr_n < 512) p.trace[w * 512 + tr_n] = (long long)__builtin_amdgcn_s_memtime(); ++tr_n; lds_barrier(); \
                       if (tr_on && tr_n < 512) p.trace[w * 512 + tr_n] = (long long)__builtin_amdgcn_s_memtime(); ++tr_n; } while (0)
#else
#define PP_SYNC() lds_barrier()
#endif
    auto idle_step = [&]() __attribute__((always_inline)) { PP_SYNC(); PP_SYNC(); PP_SYNC(); PP_SYNC(); };
    PpWin me, wprev;             // this slot's window; the one whose mel rows are still in the accumulators
    me.b = 0; me.f0 = me.f_lo = me.out_hi = me.mlen = me.valid_end = 0; me.edge = false;
    wprev = me;
    __syncthreads();
    for (int base = 0; base < n_list; base += 64) {
        unsigned long long live;
        {
            PpWin wn;
            bool lv = false;
            if (base + lane < n_list) {
                window_at(base + lane, wn);
                lv = wn.f_lo < p.L_out && wn.f_lo < wn.valid_end;
            }
            live = ballot64(lv);
            for (int k = 0; k < 64 && base + k < n_list; ++k) {      // zero-fill the windows that are all padding
                if ((live >> k) & 1ull) continue;
                window_at(base + k, wn);
                if (wn.f_lo >= p.L_out) continue;
                const int n = (wn.out_hi - wn.f_lo) * p.n_mel;
                float* o = p.mel + ((long)wn.b * p.L_out + wn.f_lo) * p.n_mel;
                for (int e = tid; e < n; e += kPpThreads) o[e] = 0.0f;
            }
        }
        const int n_live = __builtin_popcountll(live);
        const int n_mine = (n_live + 1 - slot) >> 1, n_pairs = (n_live + 1) >> 1;   // windows of this slot; trips of the window loop
        // my windows are the live ones number slot, slot + 2, ...: walk the mask
        unsigned long long rest = live;
        if (slot == 1) {
            if (rest) rest &= rest - 1;
            idle_step();
        }
        bool pend = false;       // the accumulators hold an M step's results
        for (int jw = 0; jw < n_pairs; ++jw) {
            const bool on = jw < n_mine;
            if (on) {
#ifdef ESMI_WAVESIM
                const int k = __builtin_ctzll(rest);
#else
                const int k = __builtin_amdgcn_readfirstlane(__builtin_ctzll(rest));
#endif
                rest &= rest - 1;
                if (rest) rest &= rest - 1;
                if (pend) wprev = me;
                window_at(base + k, me);
            }
            for (int l = 0; l <= n_layers; ++l) {
                if (!on) {      // (this slot has run out of windows: the last one's mel rows, then idle steps)
                    if (pend && l == 0) { mel_store(me); pend = false; }
                    idle_step();
                    idle_step();
                    continue;
                }
                // ------------------------------------------------ D step, stage 2 l (l == n_layers: the mel Linear's operand)
                {
                    const bool after_block_end = l > 0 && (l % p.block_depth) == 0;   // the M step just finished closed a decoder block
                    Strip S;
                    // quarter 0: put the previous M step's results away; a new window's rows
                    if (l == 0) {
                        if (pend) mel_store(wprev);
                        const f32x4 pv = fetch(off_A(0), GA / 4);
                        gather(me);
                        commit(pa, GA / 4, pv);
                    } else if (after_block_end) {   // u = LN(t) + skip (networks.py:299); the tile gets u, its consumer LN_s's statistics
                        float r[4], m[4];
                        merge_own(pst, r, m);
                        ln_acc(acc, pbuf + P_G, pbuf + P_B, r, m, TrueC{});
                        stats_acc(pu);
                        store_acc(me, pbuf + P_SG, pbuf + P_SB);
                    } else {
                        store_acc(me, pbuf + P_G, pbuf + P_B);
                    }
                    PP_SYNC();
                    // quarter 1: skip = LN_s(u) behind a block end; statistics of the strip's rows, halo rows
                    if (after_block_end && l < n_layers) {
                        float r[4], m[4];
                        merge_own(pu, r, m);
                        ln_acc(skip, pbuf + P_SG, pbuf + P_SB, r, m, FalseC{});
                    }
                    if (l < n_layers) cons_head(me, FalseC{}, l == 0, after_block_end ? pu : pst, S);
                    else cons_head(me, TrueC{}, false, pu, S);
                    PP_SYNC();
                    // quarters 2, 3: the strip's rows; the M step's bias / LN vectors (the old ones were last read in quarter 1) and
                    // weights, the next layer's taps (this layer's are in registers behind quarter 2's first instructions)
                    {
                        const f32x4 pv = fetch(off_B(l), l < n_layers ? GB / 4 : DX2 / 4);
                        if (l < n_layers) cons_rows(FalseC{}, I0{}, I8{}, S);
                        else cons_rows(TrueC{}, I0{}, I8{}, S);
                        commit(pbuf, l < n_layers ? GB / 4 : DX2 / 4, pv);
                    }
                    PP_SYNC();
                    {
                        const f32x4 pv = fetch(off_A(l + 1 < n_layers ? l + 1 : 0), GA / 4);
                        load_W(l < n_layers ? p.lay.layer0 + (long)l * p.lay.layer_stride + p.lay.l_pw : p.lay.mel_w);
                        if (l < n_layers) cons_rows(FalseC{}, I8{}, I16{}, S);
                        else cons_rows(TrueC{}, I8{}, I16{}, S);
                        if (l + 1 < n_layers) commit(pa, GA / 4, pv);
                    }
                    PP_SYNC();
                }
                // ------------------------------------------------ M step, stage 2 l + 1
                if (l < n_layers) {
                    k_tile(I0{}, FalseC{});
                    PP_SYNC();
                    k_tile(I1{}, TrueC{});
                    PP_SYNC();
                    k_tile(I2{}, TrueC{});
                    PP_SYNC();
                    k_tile(I3{}, TrueC{});
                    epi_tile3();
                    PP_SYNC();
                } else {
                    const bool cols = 32 * ns < p.n_mel;     // column slices beyond n_mel have nothing to do
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = zero16();
                    if (cols) k_tile(I0{}, FalseC{});
                    PP_SYNC();
                    if (cols) k_tile(I1{}, FalseC{});
                    PP_SYNC();
                    if (cols) k_tile(I2{}, FalseC{});
                    PP_SYNC();
                    if (cols) k_tile(I3{}, FalseC{});
                    PP_SYNC();
                    pend = true;
                }
            }
        }
        // the step that stores the last window's mel rows; slot 0 idles one more step so that both slots have run the same number
        if (pend) mel_store(me);
        pend = false;
        idle_step();
        if (slot == 0) idle_step();
    }
}

}  // namespace esmi
