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

namespace esmi {

struct EncAttnFfnP {
    const float* x;    // (B,N,C)   block input after the merge convs (first residual)
    const float* qkv;  // (B,N,3,h,C)
    int B, N, C, h;
    float scale;
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
__global__ __launch_bounds__(64) void enc_attn_ffn_kernel(const EncAttnFfnP p) {
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
        mx = fmaxf(mx, shfl_xor_f(mx, 32));
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
        den += shfl_xor_f(den, 32);
        const float inv = 1.0f / den;
        f32x16 o[NC];           // ctx[query][c] = sum_key P[query][key] V[key][c]
        zero_tiles<NC>(o);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r4 = 0; r4 < 16; r4 += 4) {
                float vv[4][NC];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int key = 32 * kt + tile_row(r4 + rr, lane);
                    const bool vok = key < p.N;
                    const float* vrow = vb + (long)(vok ? key : 0) * ld + i;
#pragma unroll
                    for (int nt = 0; nt < NC; ++nt) vv[rr][nt] = vok ? vrow[32 * nt] : 0.0f;
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
                    for (int nt = 0; nt < NC; ++nt) o[nt] = mfma32(s[kt][r4 + rr] * inv, vv[rr][nt], o[nt]);
                }
            }
        }
        __syncthreads();        // the previous head's proj has finished reading the tile
        tile_store<NC>(buf, LD, 0, o, lane);
        __syncthreads();
        wave_gemm<NC>(y, a_row, C, p.proj_w, p.h * C, hd * C, 0, C, lane);
    }

    // ---------------- y1 = mask(LN1(y + bias + x))
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) {
        const int col = 32 * nt + i;
        const float bc = p.proj_b[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pos = t0 + tile_row(r, lane);
            const bool ok = pos >= 0 && pos < p.N;
            y[nt][r] += bc + (ok ? p.x[((long)b * p.N + pos) * C + col] : 0.0f);
        }
    }
    layernorm_tile<NC>(y, p.ln1_g, p.ln1_b, lane);
    bool rz[16];                // rows that are padding (mask) or outside the sequence
    bool rout[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int pos = t0 + tile_row(r, lane);
        rout[r] = pos < 0 || pos >= p.N;
        rz[r] = !rout[r] && p.mask && p.mask[(long)b * p.N + pos];
    }
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (rz[r]) y[nt][r] = 0.0f;
    }
    __syncthreads();
    tile_store<NC>(buf, LD, 0, y, lane);
    __syncthreads();

    // ---------------- MixFFN: mlp1 -> dense conv k3 -> GELU -> mlp2
    f32x16 m[NE];
    zero_tiles<NE>(m);
    wave_gemm<NE>(m, a_row, C, p.mlp1_w, C, 0, 0, EC, lane);
#pragma unroll
    for (int nt = 0; nt < NE; ++nt) {
        const float bc = p.mlp1_b[32 * nt + i];
#pragma unroll
        for (int r = 0; r < 16; ++r) m[nt][r] = rout[r] ? 0.0f : m[nt][r] + bc;   // outside rows = the conv's zero padding
    }
    __syncthreads();
    tile_store<NE>(buf, LD, 0, m, lane);
    __syncthreads();
    zero_tiles<NE>(m);
#pragma unroll
    for (int j = 0; j < 3; ++j)
        wave_gemm<NE>(m, a_row + (j - 1) * LD, EC, p.conv_w + (long)j * EC * EC, EC, 0, 0, EC, lane);
#pragma unroll
    for (int nt = 0; nt < NE; ++nt) {
        const float bc = p.conv_b[32 * nt + i];
#pragma unroll
        for (int r = 0; r < 16; ++r) m[nt][r] = gelu_erf_f32(m[nt][r] + bc);
    }
    __syncthreads();
    tile_store<NE>(buf, LD, 0, m, lane);
    __syncthreads();
    f32x16 z[NC];
    zero_tiles<NC>(z);
    wave_gemm<NC>(z, a_row, EC, p.mlp2_w, EC, 0, 0, C, lane);
#pragma unroll
    for (int nt = 0; nt < NC; ++nt) {
        const float bc = p.mlp2_b[32 * nt + i];
#pragma unroll
        for (int r = 0; r < 16; ++r) z[nt][r] += bc + y[nt][r];
    }
    layernorm_tile<NC>(z, p.ln2_g, p.ln2_b, lane);
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
