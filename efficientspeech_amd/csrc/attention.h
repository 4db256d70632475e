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
    const float* qkv;  // (B, N, 3, h, C)
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
    const int ld = 3 * p.h * p.C;
    const float* base = p.qkv + (long)b * p.N * ld;
    const float* qb = base + 0 * p.h * p.C + hd * p.C;
    const float* kb = base + 1 * p.h * p.C + hd * p.C;
    const float* vb = base + 2 * p.h * p.C + hd * p.C;

    // ---- S^T[key][query] = sum_c K[key][c] Q[query][c]
    f32x16 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) s[kt] = zero16();
    const bool qok = q0 + i < p.N;
    const float* qrow = qb + (long)(qok ? q0 + i : 0) * ld;
    const float* krow[NKT];
    bool kok[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        kok[kt] = 32 * kt + i < p.N;
        krow[kt] = kb + (long)(kok[kt] ? 32 * kt + i : 0) * ld;
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
                    const float* vrow = vb + (long)(vok ? key : 0) * ld + c0 + i;
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
                    const float* vrow = vb + (long)(vok ? key : 0) * ld + c0 + i;
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
    const int ld = 3 * p.h * p.C;
    const float* base = p.qkv + (long)b * p.N * ld;
    const float* qb = base + 0 * p.h * p.C + hd * p.C;
    const float* kb = base + 1 * p.h * p.C + hd * p.C;
    const float* vb = base + 2 * p.h * p.C + hd * p.C;
    const bool qok = q0 + i < p.N;
    const float* qrow = qb + (long)(qok ? q0 + i : 0) * ld;

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
            krow[kt] = kb + (long)(kok[kt] ? k0 + 32 * kt + i : 0) * ld;
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
                    const float* vrow = vb + (long)(vok ? key : 0) * ld + i;
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

}  // namespace esmi
