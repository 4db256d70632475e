// attention core of SelfAttention.forward (layers/blocks.py:49-64):
//     attn = softmax((q @ k^T) * scale, dim=-1);  ctx = attn @ v
// for one (utterance, head, 32-query tile) per WAVE.  Faithful quirks: every head spans the FULL
// channel width C (q/k/v of head hd are channels [s*h*C + hd*C, +C) of the qkv row), scale is
// (C // heads)^-1/2 (blocks.py:37-38), and NO padding mask is applied to the scores (blocks.py:59-63
// builds one and never uses it) -- padded keys take part exactly as in the reference.
//
// Layout trick (no LDS, no transposes): the scores are computed TRANSPOSED, S^T = K Q^T, so the
// MFMA C/D registers of a lane hold one query column (query = lane&31) and 16 keys per key tile.
// Softmax over keys is then an in-lane reduction plus one xor-32 exchange, and the probabilities
// already sit in the A-operand arrangement of the second MFMA chain (ctx = P V): the k-step that
// consumes accumulator register r of key tile kt pairs key 32*kt + tile_row(r) of the low half
// wave with the same register of the high half wave, and V rows are fetched to match.
#pragma once
#include "esmi_dev.h"

namespace esmi {

struct AttnP {
    const float* qkv;  // (B, N, 3, h, C), or NULL with q / k / v given
    // the three operands on their own (launch_attn derives them from `qkv` when q is NULL): row b*N + n of head hd starts at
    // ptr + (b*N + n) * ld + hd * hs.  hs = 0 shares one tensor between the heads -- the weight-folded attention of the per-op
    // encoder plan has q = x (Wq_h^T Wk_h) per head and k = v = x for every head (esmi_encoder_block_weights.qk_w)
    const float *q, *k, *v;
    int ldq, ldk, ldv, hsq, hsk, hsv;
    int B, N, C, h;
    float scale;
    float* ctx;  // (B, N, h*C), head-major channels (blocks.py:64 transpose(1,2).reshape)
};

// NKT = key tiles of 32 (N <= 32*NKT)
template <int NKT>
__global__ __launch_bounds__(256) void attn_kernel(const AttnP p) {
    const int lane = lane_id();
    const int qtiles = (p.N + 31) >> 5;
    const int wt = (int)blockIdx.x * 4 + wave_id();
    if (wt >= p.B * p.h * qtiles) return;
    const int b = wt / (p.h * qtiles);
    const int rem = wt - b * (p.h * qtiles);
    const int hd = rem / qtiles;
    const int q0 = (rem - hd * qtiles) << 5;
    const int i = lane & 31, h2 = lane >> 5;
    const int ldq = p.ldq, ldk = p.ldk, ldv = p.ldv;
    const float* qb = p.q + (long)b * p.N * ldq + hd * p.hsq;
    const float* kb = p.k + (long)b * p.N * ldk + hd * p.hsk;
    const float* vb = p.v + (long)b * p.N * ldv + hd * p.hsv;

    // ---- S^T[key][query] = sum_c K[key][c] Q[query][c]
    f32x16 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) s[kt] = zero16();
    const bool qok = q0 + i < p.N;
    const float* qrow = qb + (long)(qok ? q0 + i : 0) * ldq;
    const float* krow[NKT];
    bool kok[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        kok[kt] = 32 * kt + i < p.N;
        krow[kt] = kb + (long)(kok[kt] ? 32 * kt + i : 0) * ldk;
    }
#if ESMI_CHAIN_SPLIT
    // split-f16x2 products (esmi_dev.h): both operands are activations, split on the fly into two binary16 pieces (no 2^8 scale:
    // |q|, |k| are far inside the binary16 range); a k-step is 16 channels, lane half h2 holds channels 16s + 8 h2 + (0..7) of its
    // row on BOTH sides.  Three v_mfma_f32_32x32x16_f16 per (key tile, 16 channels) instead of eight v_mfma_f32_32x32x2_f32.
    for (int st = 0; st < (p.C >> 4); st += 2) {       // two steps (32 channels) of operands per round trip
        f16x2p qf[2], kf[2][NKT];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int c = 16 * (st + g) + 8 * h2;
            const bool cok = st + g < (p.C >> 4);
            qf[g] = (qok && cok) ? split_f16x2(ld4(qrow + c), ld4(qrow + c + 4)) : split_f16x2(zero4(), zero4());
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
                kf[g][kt] = (kok[kt] && cok) ? split_f16x2(ld4(krow[kt] + c), ld4(krow[kt] + c + 4)) : split_f16x2(zero4(), zero4());
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                s[kt] = mfma32_f16(kf[g][kt].h2, qf[g].h1, s[kt]);
                s[kt] = mfma32_f16(kf[g][kt].h1, qf[g].h2, s[kt]);
                s[kt] = mfma32_f16(kf[g][kt].h1, qf[g].h1, s[kt]);
            }
        }
    }
#else
    // operands for 4 k-steps groups are fetched together (one memory round trip per 32 channels)
    for (int kc = 0; kc < (p.C >> 3); kc += 4) {
        f32x4 qv[4], kv[4][NKT];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 8 * (kc + g) + 4 * h2;
            qv[g] = qok ? ld4(qrow + c) : zero4();
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) kv[g][kt] = kok[kt] ? ld4(krow[kt] + c) : zero4();
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) s[kt] = mfma32(kv[g][kt][t], qv[g][t], s[kt]);
            }
        }
    }
#endif
    // ---- softmax over keys for this lane's query: in-lane over (kt, r), then the other half wave
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
            const float e = expf(s[kt][r] - mx);  // exp(-inf) = 0 for keys >= N
            s[kt][r] = e;
            den += e;
        }
    }
    den += swap32_f(den);
    const float inv = 1.0f / den;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] *= inv;
    }
    // ---- ctx[query][c] = sum_key P[query][key] V[key][c], 128 output channels per pass
    for (int c0 = 0; c0 < p.C; c0 += 128) {
        f32x16 o[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) o[nt] = zero16();
#if ESMI_CHAIN_SPLIT
        // P V on the f16 pipe: a k-step is the 16 keys the two half waves hold in accumulator registers r8 .. r8 + 7 (half h2's eight
        // k-slots are ITS eight keys, so the probabilities are already in place and every lane fetches the V rows of its own keys)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r8 = 0; r8 < 16; r8 += 8) {
                f32x4 pa, pb2;
#pragma unroll
                for (int e = 0; e < 4; ++e) { pa[e] = s[kt][r8 + e]; pb2[e] = s[kt][r8 + 4 + e]; }
                const f16x2p pf = split_f16x2(pa, pb2);
                float vv[8][4];
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int key = 32 * kt + tile_row(r8 + rr, lane);
                    const bool vok = key < p.N;
                    const float* vrow = vb + (long)(vok ? key : 0) * ldv + c0 + i;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) vv[rr][nt] = (vok && c0 + 32 * nt + i < p.C) ? vrow[32 * nt] : 0.0f;
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const f16x2p vf = split_f16x2(f32x4{vv[0][nt], vv[1][nt], vv[2][nt], vv[3][nt]}, f32x4{vv[4][nt], vv[5][nt], vv[6][nt], vv[7][nt]});
                    o[nt] = mfma32_f16(pf.h2, vf.h1, o[nt]);
                    o[nt] = mfma32_f16(pf.h1, vf.h2, o[nt]);
                    o[nt] = mfma32_f16(pf.h1, vf.h1, o[nt]);
                }
            }
        }
#else
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r4 = 0; r4 < 16; r4 += 4) {   // four key rows' V values in flight at a time
                float vv[4][4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int key = 32 * kt + tile_row(r4 + rr, lane);  // differs between the half waves: that IS the k index
                    const bool vok = key < p.N;
                    const float* vrow = vb + (long)(vok ? key : 0) * ldv + c0 + i;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) vv[rr][nt] = (vok && c0 + 32 * nt + i < p.C) ? vrow[32 * nt] : 0.0f;
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) o[nt] = mfma32(s[kt][r4 + rr], vv[rr][nt], o[nt]);
                }
            }
        }
#endif
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = q0 + tile_row(r, lane);
            if (q >= p.N) continue;
            float* orow = p.ctx + ((long)b * p.N + q) * (p.h * p.C) + hd * p.C + c0 + i;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                if (c0 + 32 * nt + i < p.C) orow[32 * nt] = o[nt][r];
        }
    }
}

// Long sequences (N > 256 keys: more than a wave can hold as one score tile set): the same transposed-score scheme over key
// CHUNKS of 128, in two sweeps -- sweep 0 finds each query's row maximum and normaliser (running max / rescaled sum, per-lane
// scalars only), sweep 1 recomputes the scores, forms P = exp(s - max) / sum exactly as the one-chunk kernel does and
// accumulates ctx = P V for all C channels (NC = C/32 accumulator tiles).  Twice the Q K^T work buys a kernel with no
// cross-lane rescale of the output tiles; the reference has no sequence limit (README "long text"), this is its path here.
template <int NC>
__global__ __launch_bounds__(256) void attn_long_kernel(const AttnP p) {
    constexpr int NKT = 4;
    const int lane = lane_id();
    const int qtiles = (p.N + 31) >> 5;
    const int wt = (int)blockIdx.x * 4 + wave_id();
    if (wt >= p.B * p.h * qtiles) return;
    const int b = wt / (p.h * qtiles);
    const int rem = wt - b * (p.h * qtiles);
    const int hd = rem / qtiles;
    const int q0 = (rem - hd * qtiles) << 5;
    const int i = lane & 31, h2 = lane >> 5;
    const int ldq = p.ldq, ldk = p.ldk, ldv = p.ldv;
    const float* qb = p.q + (long)b * p.N * ldq + hd * p.hsq;
    const float* kb = p.k + (long)b * p.N * ldk + hd * p.hsk;
    const float* vb = p.v + (long)b * p.N * ldv + hd * p.hsv;
    const bool qok = q0 + i < p.N;
    const float* qrow = qb + (long)(qok ? q0 + i : 0) * ldq;

    f32x16 s[NKT];
    // scores of key chunk [k0, k0 + 128) for this lane's query, scaled; keys >= N -> -inf
    auto scores = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) s[kt] = zero16();
        const float* krow[NKT];
        bool kok[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            kok[kt] = k0 + 32 * kt + i < p.N;
            krow[kt] = kb + (long)(kok[kt] ? k0 + 32 * kt + i : 0) * ldk;
        }
        for (int kc = 0; kc < (p.C >> 3); kc += 4) {
            f32x4 qv[4], kv[4][NKT];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 8 * (kc + g) + 4 * h2;
                qv[g] = qok ? ld4(qrow + c) : zero4();
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) kv[g][kt] = kok[kt] ? ld4(krow[kt] + c) : zero4();
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) s[kt] = mfma32(kv[g][kt][t], qv[g][t], s[kt]);
                }
            }
        }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + 32 * kt + tile_row(r, lane);
                s[kt][r] = key < p.N ? s[kt][r] * p.scale : -INFINITY;
            }
        }
    };
    // ---- sweep 0: row maximum and normaliser
    float mx = -INFINITY, den = 0.0f;
    for (int k0 = 0; k0 < p.N; k0 += 32 * NKT) {
        scores(k0);
        float cm = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) cm = fmaxf(cm, s[kt][r]);
        }
        cm = fmaxf(cm, swap32_f(cm));
        const float nm = fmaxf(mx, cm);            // finite from the first chunk on (it holds at least one real key)
        float cs = 0.0f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) cs += expf(s[kt][r] - nm);
        }
        cs += swap32_f(cs);
        den = den * expf(mx - nm) + cs;            // exp(-inf) = 0 on the first chunk
        mx = nm;
    }
    const float inv = 1.0f / den;
    // ---- sweep 1: P and ctx = P V
    f32x16 o[NC];
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) o[nt] = zero16();
    for (int k0 = 0; k0 < p.N; k0 += 32 * NKT) {
        scores(k0);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = expf(s[kt][r] - mx) * inv;
        }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r4 = 0; r4 < 16; r4 += 4) {
                float vv[4][NC];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int key = k0 + 32 * kt + tile_row(r4 + rr, lane);
                    const bool vok = key < p.N;
                    const float* vrow = vb + (long)(vok ? key : 0) * ldv + i;
#pragma unroll
                    for (int nt = 0; nt < NC; ++nt) vv[rr][nt] = vok ? vrow[32 * nt] : 0.0f;
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
                    for (int nt = 0; nt < NC; ++nt) o[nt] = mfma32(s[kt][r4 + rr], vv[rr][nt], o[nt]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int q = q0 + tile_row(r, lane);
        if (q >= p.N) continue;
        float* orow = p.ctx + ((long)b * p.N + q) * (p.h * p.C) + hd * p.C + i;
#pragma unroll
        for (int nt = 0; nt < NC; ++nt) orow[32 * nt] = o[nt][r];
    }
}

#if ESMI_CHAIN_SPLIT
// ---- the same attention with K, then V, staged ONCE per (utterance, head) in LDS as split binary16 planes.  attn_kernel
// streams K and V from L2 for every 32-query tile (base ES block 0: 8 tiles per head -> 4 GB of L2 reads per call, which is
// what bounds it); here one workgroup of NKT waves (= all query tiles of the head, N <= 32 NKT <= 256) shares them:
//   stage K planes [key][C] -> scores S^T = K Q^T (A fragments: two ds_read_b128 per key tile and 16 channels, no per-use
//   split; Q split once per wave and step) -> softmax in registers, as attn_kernel -> barrier -> stage V TRANSPOSED into the
//   same LDS, Vt[channel][k-slot], with the k-slots permuted so that the eight keys a half wave contributes to a 16-key step
//   are contiguous -> ctx = P V (B fragments: two ds_read_b128 per 32 channels and step).
// Row strides 2 C + 16 / 2 N + 16 bytes: the 16-byte fragment reads of 32 consecutive rows are bank-conflict free.
__host__ __device__ inline size_t attn_lds_bytes(int N, int C) {
    const int nk = N <= 128 ? 128 : 256;                  // the two instantiations stage 32 NKT = 128 or 256 key rows
    const int ck = C < 128 ? C : 128;                     // ... 128 channels at a time (K: per pass of the score loop, V: per output pass)
    const size_t k = (size_t)2 * nk * (2 * ck + 16), v = (size_t)2 * ck * (2 * nk + 16);
    return k > v ? k : v;
}
#ifndef ESMI_ATTN_KO
#define ESMI_ATTN_KO 0   // development (tools/probes/probe_attn.hip): knock a phase out to see what it costs -- 1 K writes, 2 score products,
#endif                   // 4 V writes, 8 P V products, 16 softmax, 32 K / V requests, 64 Q requests.  0 in every library build.
// Round 6: every global read is issued a phase ahead of its use.  The round-5 form ran `load -> convert -> ds_write` once per item in
// runtime-bounded loops (16 dependent L2 round trips per K pass and thread) and fetched the Q fragments inside the product loop, at two
// waves per SIMD: base ES block 1 (N = 128, C = 256) took 260 us for 17 GFLOP.  Now a phase's items (K: CK / 8 float4 per thread, V: CK / 32
// groups of four) are requested together into `P` -- the next phase's while this phase's products run (NKT <= 4: there are registers for
// it), or right behind the score loop in front of the softmax (NKT = 8) -- and a pass's Q fragments are requested in two batches, the
// first in front of the staging, the second in front of the first batch's products.
template <int NKT, int CK>      // CK = min(C, 128): 32, 64 or 128
__global__ __launch_bounds__(64 * NKT, NKT <= 4 ? 2 : 1) void attn_lds_kernel(const AttnP p) {
    ESMI_DYN_LDS(lds_f);
    char* lds = reinterpret_cast<char*>(lds_f);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    constexpr int NK = 32 * NKT, NTHR = 64 * NKT;
    constexpr bool kAhead = NKT <= 4;                         // the next phase's rows are in flight under this phase's products
    const int lane = lane_id(), w = wave_id(), tid = (int)threadIdx.x;
    const int hd = (int)blockIdx.x % p.h, b = (int)blockIdx.x / p.h;
    const int i = lane & 31, h2 = lane >> 5, q0 = 32 * w;
    const int ldq = p.ldq, ldk = p.ldk, ldv = p.ldv, C = p.C;
    // buffer resources over this (utterance, head)'s rows: keys / queries >= N read as zeros, and the wave-uniform part of every offset
    // (item index, pass, row of a group) travels in an SGPR -- one lane-offset register per operand instead of a 64-bit pointer per item
    const BufRsrc r_q = make_rsrc(p.q + (long)b * p.N * ldq + hd * p.hsq, ((long)(p.N - 1) * ldq + C) * 4);
    const BufRsrc r_k = make_rsrc(p.k + (long)b * p.N * ldk + hd * p.hsk, ((long)(p.N - 1) * ldk + C) * 4);
    const BufRsrc r_v = make_rsrc(p.v + (long)b * p.N * ldv + hd * p.hsv, ((long)(p.N - 1) * ldv + C) * 4);
    const BufRsrc r_o = make_rsrc(p.ctx + (long)b * p.N * (p.h * C) + hd * C, ((long)(p.N - 1) * (p.h * C) + C) * 4);
    const unsigned o_off = (unsigned)(((q0 + 4 * h2) * (p.h * C) + i) * 4);   // (row q0 + tile_row(0, lane), column i of the head's channels)
    constexpr int krs = 2 * CK + 16, kplane = NK * krs;       // K planes: [2][NK][krs bytes], CK channels per pass
    constexpr int vrs = 2 * NK + 16, vplane = CK * vrs;       // Vt planes: [2][CK][vrs bytes]
    constexpr int c4sh = CK == 128 ? 5 : (CK == 64 ? 4 : 3);  // log2 of the float4 items per key row and pass
    constexpr int nK = CK >> 3;                               // K items (4 channels of one key) per thread and pass: NK (CK / 4) / NTHR
    constexpr int nV = CK >> 5;                               // V items (4 keys x 4 channels) per thread and pass: (NK / 4) (CK / 4) / NTHR
    constexpr int rpi = NTHR >> c4sh;                         // item u of a thread is rpi rows (K) / row groups (V) below item u - 1
    static_assert(rpi % 8 == 0, "attn_lds: a thread's V items are whole key tiles apart");
    const int npass = C / CK;
    const int e_row = tid >> c4sh, e_c = (tid - (e_row << c4sh)) << 2;   // this thread's item 0: row (K) or group of four rows (V), first channel
    const unsigned k_off = (unsigned)((e_row * ldk + e_c) * 4), v_off = (unsigned)((4 * e_row * ldv + e_c) * 4);
    f32x4 P[16];
    auto request_k = [&](int cp) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < nK; ++u) P[u] = (ESMI_ATTN_KO & 32) ? zero4() : buf_ld4s(r_k, k_off, (unsigned)((u * rpi * ldk + cp) * 4));
    };
    auto stage_k = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < nK; ++u) {
            unsigned h1a, h2a, h1b, h2b;
            split_f16_pair(P[u][0], P[u][1], h1a, h2a);
            split_f16_pair(P[u][2], P[u][3], h1b, h2b);
            char* d = lds + opaque_i(e_row * krs + e_c * 2);        // (+ compile-time offsets: nothing per item is kept in registers)
            if ((ESMI_ATTN_KO & 1) && h1a != 12345u) continue;
            *reinterpret_cast<u32x2*>(d + u * rpi * krs) = u32x2{h1a, h1b};
            *reinterpret_cast<u32x2*>(d + kplane + u * rpi * krs) = u32x2{h2a, h2b};
        }
    };
    // V is staged TRANSPOSED: one item = 4 consecutive keys x 4 channels; k-slot of key 32 kt + 8 q + 4 h + e (e < 4) is
    // 32 kt + 16 (q >> 1) + 8 h + 4 (q & 1) + e, i.e. the eight keys of (kt, step q >> 1, half h) are slots 8 h .. 8 h + 7 of that 16-key step
    auto request_v = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < nV; ++u) {
#pragma unroll
            for (int r = 0; r < 4; ++r) P[4 * u + r] = (ESMI_ATTN_KO & 32) ? zero4() : buf_ld4s(r_v, v_off, (unsigned)(((4 * u * rpi + r) * ldv + c0) * 4));
        }
    };
    auto stage_v = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < nV; ++u) {
            // item u's keys are 4 rpi u below item 0's (a multiple of 32: same q, h; kt + rpi u / 8), i.e. 4 rpi u slots further
            const int key0 = 4 * e_row, kt = key0 >> 5, q = (key0 >> 3) & 3, hh = (key0 >> 2) & 1;
            const int slot0 = 32 * kt + 16 * (q >> 1) + 8 * hh + 4 * (q & 1);
            char* d = lds + opaque_i(e_c * vrs + slot0 * 2);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                unsigned h1a, h2a, h1b, h2b;
                split_f16_pair(P[4 * u][cc], P[4 * u + 1][cc], h1a, h2a);
                split_f16_pair(P[4 * u + 2][cc], P[4 * u + 3][cc], h1b, h2b);
                if ((ESMI_ATTN_KO & 4) && h1a != 12345u) continue;
                *reinterpret_cast<u32x2*>(d + cc * vrs + 8 * rpi * u) = u32x2{h1a, h1b};
                *reinterpret_cast<u32x2*>(d + vplane + cc * vrs + 8 * rpi * u) = u32x2{h2a, h2b};
            }
        }
    };

    // ---- S^T[key][query] for this wave's 32 queries, CK channels of K staged at a time
    f32x16 s[NKT];                                            // (zeroed behind pass 0's barrier: 32 NKT registers the first requests can use)
    const unsigned q_off = q0 + i < p.N && !(ESMI_ATTN_KO & 64) ? (unsigned)(((q0 + i) * ldq + 8 * h2) * 4) : kBufOOB;
    constexpr int nst = CK >> 4;                              // 16-channel steps per pass: 2, 4 or 8
    if (kAhead) request_k(0);
    for (int pass = 0; pass < npass; ++pass) {
        const int cp = pass * CK;
        f32x4 qa[4][2], qb2[4][2];
        if (!kAhead) request_k(cp);                           // (NKT = 8: s is half the register file, nothing is carried around the loop)
#pragma unroll
        for (int t = 0; t < 4; ++t) {                         // steps 0 .. 3 of the pass: on their way while K is staged
            if (t < nst) {
                qa[t][0] = buf_ld4s(r_q, q_off, (unsigned)((cp + 16 * t) * 4));
                qa[t][1] = buf_ld4s(r_q, q_off, (unsigned)((cp + 16 * t + 4) * 4));
            }
        }
        if (pass) __syncthreads();                            // the previous pass's fragments have been read
        stage_k();
        if (kAhead) {
            if (pass + 1 < npass) request_k(cp + CK); else request_v(0);
        }
        __syncthreads();
        if (pass == 0) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) s[kt] = zero16();
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {                         // steps 4 .. 7: on their way under the products of steps 0 .. 3
            if (4 + t < nst) {
                qb2[t][0] = buf_ld4s(r_q, q_off, (unsigned)((cp + 16 * (4 + t)) * 4));
                qb2[t][1] = buf_ld4s(r_q, q_off, (unsigned)((cp + 16 * (4 + t) + 4) * 4));
            }
        }
        // fragment addresses: two lane-dependent bases (one per plane: kplane is beyond the 16-bit DS offset field) + compile-time offsets
        const char* const ka1 = lds + opaque_i(i * krs + 16 * h2);
        const char* const ka2 = lds + opaque_i(i * krs + 16 * h2 + kplane);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int st = 4 * half + t;
                if (st < nst) {
                    const f16x2p qf = half ? split_f16x2(qb2[t][0], qb2[t][1]) : split_f16x2(qa[t][0], qa[t][1]);
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) {
                        if (ESMI_ATTN_KO & 2) { s[kt][st] += __builtin_bit_cast(float, qf.h1[0] ^ qf.h2[1]); continue; }
                        const u32x4 k1 = *reinterpret_cast<const u32x4*>(ka1 + 32 * kt * krs + 32 * st);
                        const u32x4 k2 = *reinterpret_cast<const u32x4*>(ka2 + 32 * kt * krs + 32 * st);
                        s[kt] = mfma32_f16(k2, qf.h1, s[kt]);
                        s[kt] = mfma32_f16(k1, qf.h2, s[kt]);
                        s[kt] = mfma32_f16(k1, qf.h1, s[kt]);
                    }
                }
            }
        }
    }
    if (!kAhead) request_v(0);
    // ---- softmax over keys (attn_kernel's); the first V pass's rows are in flight
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
            const float e = (ESMI_ATTN_KO & 16) ? s[kt][r] - mx : expf(s[kt][r] - mx);
            s[kt][r] = e;
            den += e;
        }
    }
    den += swap32_f(den);
    const float inv = 1.0f / den;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] *= inv;
    }

    // ---- ctx[query][c] = sum_key P[query][key] V[key][c], CK output channels per pass
    for (int vp = 0; vp < npass; ++vp) {
        const int c0 = vp * CK;
        if (!kAhead && vp) request_v(c0);
        __syncthreads();                                   // K (or the previous pass's V) has been read by every wave
        stage_v();
        if (kAhead && vp + 1 < npass) request_v(c0 + CK);
        __syncthreads();
        // 64 output channels at a time (round 5: with all 128 in flight -- s: 128 registers, o: 64 -- the <8> instantiation spilled 228 B per
        // lane; the probabilities are split again for the second half, eight conversions per step against twelve MFMAs)
        for (int nh = 0; 64 * nh < CK; ++nh) {
            f32x16 o[2] = {zero16(), zero16()};
            const char* const va1 = lds + opaque_i((64 * nh + i) * vrs + 16 * h2);
            const char* const va2 = lds + opaque_i((64 * nh + i) * vrs + 16 * h2 + vplane);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int r8 = 0; r8 < 16; r8 += 8) {
                    f32x4 pa, pb2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { pa[e] = s[kt][r8 + e]; pb2[e] = s[kt][r8 + 4 + e]; }
                    const f16x2p pf = split_f16x2(pa, pb2);
                    // (k-slots 32 kt + 2 r8 + 8 h2 + (0..7): this half wave's eight of the step)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        if (32 * (2 * nh + nt) >= CK) continue;           // workgroup-uniform
                        if (ESMI_ATTN_KO & 8) { o[nt][kt] += __builtin_bit_cast(float, pf.h1[0] ^ pf.h2[1]); continue; }
                        const u32x4 v1 = *reinterpret_cast<const u32x4*>(va1 + 32 * nt * vrs + (32 * kt + 2 * r8) * 2);
                        const u32x4 v2 = *reinterpret_cast<const u32x4*>(va2 + 32 * nt * vrs + (32 * kt + 2 * r8) * 2);
                        o[nt] = mfma32_f16(pf.h2, v1, o[nt]);
                        o[nt] = mfma32_f16(pf.h1, v2, o[nt]);
                        o[nt] = mfma32_f16(pf.h1, v1, o[nt]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {                 // row q0 + tile_row(r, lane) = this lane's first row + (r & 3) + 8 (r >> 2); rows >= N are out of range
                const unsigned so = (unsigned)((((r & 3) + 8 * (r >> 2)) * (p.h * C) + c0 + 64 * nh) * 4);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    if (64 * nh + 32 * nt < CK) buf_st_s(r_o, o_off, so + 128u * nt, o[nt][r]);
            }
        }
    }
}
#endif

}  // namespace esmi
