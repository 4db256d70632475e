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
    int wgs_per_b;                  // workgroups per utterance
    int useful;                     // positions stored per workgroup: 32*nw - 2*halo
    int halo;                       // 0: one workgroup covers the sequence, 1: one recomputed row per side
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

template <int NKT, int NC, int E>   // keys <= 32*NKT, C = 32*NC, MixFFN hidden = E*C
__global__ __launch_bounds__(64 * kEncMaxWaves, ESMI_E2_WPS) void enc_attn_ffn_kernel(const EncAttnFfnP p) {
    constexpr int NE = NC * E;
    constexpr int C = 32 * NC, EC = 32 * NE;
    constexpr int LD = EC + 4;
    ESMI_DYN_LDS(lds);              // [32*nw + 2][LD]: first and last row are the zero rows around the workgroup's tile
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
    const BufRsrc r_mask = make_rsrc(p.mask ? p.mask + (long)b * p.N : nullptr, p.N);
    const BufRsrc r_out = make_rsrc(p.out + (long)b * p.N * C, (long)p.N * C * 4);
    ESMI_CT_INIT(NC == 1 ? 0 : 1);
    ESMI_CT();   // 0 start

    // ---------------- prologue: everything that does not depend on a result is requested now, nothing is waited for.
    // Rows outside [0, N) are out of range of their buffers: negative positions wrap to huge unsigned offsets.
    const unsigned mb = buf_ld_u8(r_mask, (unsigned)pos_i);
    const unsigned q_off = (unsigned)((pos_i * ld + 4 * h2) * 4);
    unsigned k_off[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) k_off[kt] = (unsigned)((((32 * kt + i) * ld) + p.h * C + 4 * h2) * 4);
    WaveGrp<NC> gp;                 // proj weights, head 0
    wave_prefetch<NC>(gp, p.proj_w, NC, 0, 0, lane);
    constexpr bool kHoistRes = NC <= 2;
    f32x16 xres[kHoistRes ? NC : 1];
    if (kHoistRes) {
#pragma unroll
        for (int nt = 0; nt < (kHoistRes ? NC : 1); ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                xres[nt][r] = buf_ld(r_x, (unsigned)(((t0 + tile_row(r, lane)) * C + 32 * nt + i) * 4));
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
        const unsigned hd_off = (unsigned)(hd * C * 4);
        const unsigned v_base = (unsigned)(((2 * p.h + hd) * C + i) * 4);
        f32x16 s[NKT];
        zero_tiles<NKT>(s);
        for (int kc = 0; kc < (C >> 3); kc += 4) {   // S^T[key][query] = sum_c K[key][c] Q[query][c]
            f32x4 qv[4], kv[4][NKT];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                qv[g] = buf_ld4(r_qkv, q_off + hd_off + 32u * (kc + g));
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) kv[g][kt] = buf_ld4(r_qkv, k_off[kt] + hd_off + 32u * (kc + g));
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
        // P V operands: 4*NKT groups of four key rows; the first two groups are requested before the softmax.
        // keys >= N: V reads 0 (buffer bound) and P = 0.
        struct VG { float v[4][NC]; };
        auto vfetch = [&](int f, VG& gq) __attribute__((always_inline)) {
            const int kt = f >> 2, r4 = (f & 3) << 2;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int key = 32 * kt + tile_row(r4 + rr, lane);   // differs between the half waves: that IS the k index
#pragma unroll
                for (int nt = 0; nt < NC; ++nt) gq.v[rr][nt] = buf_ld(r_qkv, v_base + (unsigned)((key * ld + 32 * nt) * 4));
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
#pragma unroll
        for (int f = 0; f < 4 * NKT; f += 2) {
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
            if (f + 3 < 4 * NKT) vfetch(f + 3, v1);
        }
        ESMI_CT();   // 3 PV issued
        lds_wave_sync();        // the previous head's proj has finished reading the tile
        tile_store<NC>(buf, LD, 0, o, lane);
        lds_wave_sync();
        wave_gemm<NC>(y, gp, a_row, true, C, p.proj_w, NC, (hd * C) >> 3, 0, lane);
        if (hd + 1 < p.h) wave_prefetch<NC>(gp, p.proj_w, NC, ((hd + 1) * C) >> 3, 0, lane);
    }
    WaveGrp<NE> gm;                 // mlp1 weights
    wave_prefetch<NE>(gm, p.mlp1_w, NE, 0, 0, lane);

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
            if (kHoistRes) xr = xres[kHoistRes ? nt : 0][r];
            else xr = buf_ld(r_x, (unsigned)(((t0 + tile_row(r, lane)) * C + col) * 4));
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
    lds_wave_sync();
    tile_store<NC>(buf, LD, 0, y, lane);
    lds_wave_sync();

    ESMI_CT();   // 5 LN1 + store done
    // ---------------- MixFFN: mlp1 -> dense conv k3 -> GELU -> mlp2
    f32x16 m[NE];
    zero_tiles<NE>(m);
    wave_gemm<NE>(m, gm, a_row, true, C, p.mlp1_w, NE, 0, 0, lane);
    wave_prefetch<NE>(gm, p.conv_w, NE, 0, 0, lane);   // conv weights, tap 0
#pragma unroll
    for (int nt = 0; nt < NE; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) m[nt][r] = rout[r] ? 0.0f : m[nt][r] + m1b_[nt];   // outside rows = the conv's zero padding
    }
    lds_wave_sync();
    tile_store<NE>(buf, LD, 0, m, lane);
    __syncthreads();            // the neighbouring waves' boundary rows (and the zero rows) are in place
    ESMI_CT();   // 6 mlp1 + store
    zero_tiles<NE>(m);
    {
        const float* const taps[3] = {a_row - LD, a_row, a_row + LD};
        const bool tok[3] = {true, true, true};
        wave_gemm_taps<NE, 3, NE, false>(m, gm, taps, tok, 3, p.conv_w, (long)EC * EC, NE, 0, 0, lane);
    }
    WaveGrp<NC> g2;                 // mlp2 weights
    wave_prefetch<NC>(g2, p.mlp2_w, NC, 0, 0, lane);
#pragma unroll
    for (int nt = 0; nt < NE; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) m[nt][r] = gelu_erf_f32(m[nt][r] + cb_[nt]);
    }
    __syncthreads();            // every wave has read its neighbours' rows
    tile_store<NE>(buf, LD, 0, m, lane);
    lds_wave_sync();
    ESMI_CT();   // 7 conv + gelu + store
    f32x16 z[NC];
    zero_tiles<NC>(z);
    wave_gemm<NC>(z, g2, a_row, true, EC, p.mlp2_w, NC, 0, 0, lane);
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

}  // namespace esmi
