// Fused second half of an encoder block (layers/networks.py:72-85, layers/blocks.py:22-29,49-66):
//
//     y  = proj(softmax(q k^T * scale) v) + bias            SelfAttention (scores NOT masked, heads full width)
//     y1 = mask(LN1(y + x))                                  networks.py:73-75
//     f  = mask(LN2(mlp2(GELU(conv3(mlp1(y1)))) + y1))       MixFFN + networks.py:80-83
//
// one wave per (utterance, 30-position tile): the tile carries one halo row on each side because the dense
// k=3 conv inside MixFFN needs its neighbours' mlp1 outputs; those two rows are recomputed (attention
// included) instead of exchanged.  Six kernel launches and five HBM round trips of the unfused path become
// one launch; intermediates live in registers and in a private LDS tile [34][hidden+4].
#pragma once
#include "wave_chain.h"

#ifndef ESMI_E2_WPS
#define ESMI_E2_WPS ESMI_CHAIN_WPS
#endif

namespace esmi {

struct EncAttnFfnP {
    const float* x;    // (B,N,C)   block input after the merge convs (first residual)
    const float* qkv;  // (B,N,3,h,C)
    int B, N, C, h;
    float scale;
    // the four weight matrices are in MFMA B-fragment order (esmi_pack_bfrag_f32, see wave_chain.h)
    const float *proj_w, *proj_b;   // (C, h*C), (C)
    const float *ln1_g, *ln1_b;
    const float *mlp1_w, *mlp1_b;   // (E*C, C)
    const float *conv_w, *conv_b;   // (3, E*C, E*C) tap-major
    const float *mlp2_w, *mlp2_b;   // (C, E*C)
    const float *ln2_g, *ln2_b;
    const unsigned char* mask;      // (B,N) or NULL
    float* out;                     // (B,N,C)
    int tiles_per_b;                // ceil(N / 30)
};

constexpr int kEncTileRows = 30;    // useful rows per 32-row tile (one halo row each side)

template <int NKT, int NC, int E>   // keys <= 32*NKT, C = 32*NC, MixFFN hidden = E*C
__global__ __launch_bounds__(64, ESMI_E2_WPS) void enc_attn_ffn_kernel(const EncAttnFfnP p) {
    constexpr int NE = NC * E;
    constexpr int C = 32 * NC, EC = 32 * NE;
    constexpr int LD = EC + 4;
    ESMI_DYN_LDS(lds);              // [34][LD]: row 0 and row 33 are the zero rows around the 32-row tile
    float* buf = lds + LD;
    const int lane = lane_id(), i = lane & 31, h2 = lane >> 5;
    const int b = (int)blockIdx.x / p.tiles_per_b, tile = (int)blockIdx.x - b * p.tiles_per_b;
    const int t0 = tile * kEncTileRows - 1;   // sequence position of tile row 0
    for (int c = lane; c < LD; c += 64) {
        lds[c] = 0.0f;
        lds[33 * LD + c] = 0.0f;
    }
    const int pos_i = t0 + i;
    const bool in_i = pos_i >= 0 && pos_i < p.N;
    const float* a_row = buf + i * LD + 4 * h2;
    const int ld = 3 * p.h * C;
    const float* base = p.qkv + (long)b * p.N * ld;
    ESMI_CT_INIT(NC == 1 ? 0 : 1);
    ESMI_CT();   // 0 start
    // Everything that does not depend on the attention result is requested now, so that its memory round trips
    // (~2 us each for tensors the previous kernel just wrote) overlap the attention instead of following it.
    bool rz[16], rout[16];      // rows that are padding (mask) / outside the sequence
    {   // ONE mask byte per lane (row i) + a ballot, instead of 16 dependent byte loads per lane
        const unsigned char mb = (in_i && p.mask) ? p.mask[(long)b * p.N + pos_i] : (unsigned char)0;
        const unsigned mbits = (unsigned)ballot64(mb != 0);   // bit i = row i (both half waves hold the same rows)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tile_row(r, lane);
            const int pos = t0 + row;
            rout[r] = pos < 0 || pos >= p.N;
            rz[r] = !rout[r] && ((mbits >> row) & 1u);
        }
    }
    constexpr bool kHoistRes = NC <= 2;
    f32x16 xres[kHoistRes ? NC : 1];
    if (kHoistRes) {
#pragma unroll
        for (int nt = 0; nt < (kHoistRes ? NC : 1); ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                xres[nt][r] = rout[r] ? 0.0f : p.x[((long)b * p.N + t0 + tile_row(r, lane)) * C + 32 * nt + i];
        }
    }
    float pb_[NC], g1_[NC], be1_[NC], b2_[NC], g2_[NC], be2_[NC], m1b_[NE], cb_[NE];
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) {
        const int col = 32 * nt + i;
        pb_[nt] = p.proj_b[col]; g1_[nt] = p.ln1_g[col]; be1_[nt] = p.ln1_b[col];
        b2_[nt] = p.mlp2_b[col]; g2_[nt] = p.ln2_g[col]; be2_[nt] = p.ln2_b[col];
    }
#pragma unroll
    for (int nt = 0; nt < NE; ++nt) {
        m1b_[nt] = p.mlp1_b[32 * nt + i];
        cb_[nt] = p.conv_b[32 * nt + i];
    }

    // ---------------- attention, one head at a time; proj accumulates over heads
    f32x16 y[NC];
    zero_tiles<NC>(y);
    for (int hd = 0; hd < p.h; ++hd) {
        const float* qb = base + 0 * p.h * C + hd * C;
        const float* kb = base + 1 * p.h * C + hd * C;
        const float* vb = base + 2 * p.h * C + hd * C;
        f32x16 s[NKT];
        zero_tiles<NKT>(s);
        const float* qrow = qb + (long)(in_i ? pos_i : 0) * ld + 4 * h2;
        const float* krow[NKT];
        bool kok[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            kok[kt] = 32 * kt + i < p.N;
            krow[kt] = kb + (long)(kok[kt] ? 32 * kt + i : 0) * ld + 4 * h2;
        }
        for (int kc = 0; kc < (C >> 3); kc += 4) {   // S^T[key][query] = sum_c K[key][c] Q[query][c]
            f32x4 qv[4], kv[4][NKT];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                qv[g] = in_i ? ld4(qrow + 8 * (kc + g)) : zero4();
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) kv[g][kt] = kok[kt] ? ld4(krow[kt] + 8 * (kc + g)) : zero4();
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
        ESMI_CT();   // 1 S^T done
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
                const float e = expf(s[kt][r] - mx);
                s[kt][r] = e;
                den += e;
            }
        }
        den += swap32_f(den);
        const float inv = 1.0f / den;
        ESMI_CT();   // 2 softmax done
        f32x16 o[NC];           // ctx[query][c] = sum_key P[query][key] V[key][c]
        zero_tiles<NC>(o);
        {   // P V: 4*NKT groups of four key rows, software pipelined (next group's V rows in flight during the MFMAs)
            struct VG { float v[4][NC]; };
            auto vfetch = [&](int f, VG& gq) __attribute__((always_inline)) {
                const int kt = f >> 2, r4 = (f & 3) << 2;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int key = 32 * kt + tile_row(r4 + rr, lane);   // differs between the half waves: that IS the k index
                    const bool vok = key < p.N;
                    const float* vrow = vb + (long)(vok ? key : 0) * ld + i;
#pragma unroll
                    for (int nt = 0; nt < NC; ++nt) gq.v[rr][nt] = vok ? vrow[32 * nt] : 0.0f;
                }
            };
            VG v0, v1;
            vfetch(0, v0);
#pragma unroll
            for (int f = 0; f < 4 * NKT; f += 2) {
                vfetch(f + 1, v1);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
                    for (int nt = 0; nt < NC; ++nt)
                        o[nt] = mfma32(s[f >> 2][((f & 3) << 2) + rr] * inv, v0.v[rr][nt], o[nt]);
                }
                if (f + 2 < 4 * NKT) vfetch(f + 2, v0);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
                    for (int nt = 0; nt < NC; ++nt)
                        o[nt] = mfma32(s[(f + 1) >> 2][(((f + 1) & 3) << 2) + rr] * inv, v1.v[rr][nt], o[nt]);
                }
            }
        }
        ESMI_CT();   // 3 PV done
        __syncthreads();        // the previous head's proj has finished reading the tile
        tile_store<NC>(buf, LD, 0, o, lane);
        __syncthreads();
        wave_gemm<NC>(y, a_row, true, C, p.proj_w, NC, (hd * C) >> 3, 0, lane);
    }

    ESMI_CT();   // 4 proj done
    // ---------------- y1 = mask(LN1(y + bias + x))
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) {
        const int col = 32 * nt + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float xr;
            if (kHoistRes) xr = xres[kHoistRes ? nt : 0][r];
            else xr = rout[r] ? 0.0f : p.x[((long)b * p.N + t0 + tile_row(r, lane)) * C + col];
            y[nt][r] += pb_[nt] + xr;
        }
    }
    layernorm_tile_regs<NC>(y, g1_, be1_);
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (rz[r]) y[nt][r] = 0.0f;
    }
    __syncthreads();
    tile_store<NC>(buf, LD, 0, y, lane);
    __syncthreads();

    ESMI_CT();   // 5 LN1 + store done
    // ---------------- MixFFN: mlp1 -> dense conv k3 -> GELU -> mlp2
    f32x16 m[NE];
    zero_tiles<NE>(m);
    wave_gemm<NE>(m, a_row, true, C, p.mlp1_w, NE, 0, 0, lane);
#pragma unroll
    for (int nt = 0; nt < NE; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) m[nt][r] = rout[r] ? 0.0f : m[nt][r] + m1b_[nt];   // outside rows = the conv's zero padding
    }
    __syncthreads();
    tile_store<NE>(buf, LD, 0, m, lane);
    __syncthreads();
    ESMI_CT();   // 6 mlp1 + store
    zero_tiles<NE>(m);
    {
        const float* const taps[3] = {a_row - LD, a_row, a_row + LD};
        const bool tok[3] = {true, true, true};   // rows 0 and 33 of the tile are the zero rows
        wave_gemm_taps<NE, 3, NE, false>(m, taps, tok, 3, p.conv_w, (long)EC * EC, NE, 0, 0, lane);
    }
#pragma unroll
    for (int nt = 0; nt < NE; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) m[nt][r] = gelu_erf_f32(m[nt][r] + cb_[nt]);
    }
    __syncthreads();
    tile_store<NE>(buf, LD, 0, m, lane);
    __syncthreads();
    ESMI_CT();   // 7 conv + gelu + store
    f32x16 z[NC];
    zero_tiles<NC>(z);
    wave_gemm<NC>(z, a_row, true, EC, p.mlp2_w, NC, 0, 0, lane);
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) z[nt][r] += b2_[nt] + y[nt][r];
    }
    ESMI_CT();   // 8 mlp2
    layernorm_tile_regs<NC>(z, g2_, be2_);
    ESMI_CT();   // 9 LN2
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = tile_row(r, lane);
        if (row < 1 || row > kEncTileRows || rout[r]) continue;   // halo rows / beyond the sequence end
        float* orow = p.out + ((long)b * p.N + t0 + row) * C + i;
#pragma unroll
        for (int nt = 0; nt < NC; ++nt) orow[32 * nt] = rz[r] ? 0.0f : z[nt][r];
    }
}

}  // namespace esmi
