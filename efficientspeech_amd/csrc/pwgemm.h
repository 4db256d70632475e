// pwgemm -- a Linear / k = 1 convolution over MANY rows with the whole weight resident in LDS (the training step's decoder
// GEMMs: 76,800 rows x 128 -> 128 channels at B = 128, forward and data gradient; networks.py:151-160 / model.py:212-226).
//
// convgemm_dma_kernel (convgemm.h) gives a 4-wave workgroup 128 rows and walks the contraction in 32-channel chunks: weights of a
// chunk fetched, split and staged by the workgroup, one barrier per chunk.  With K = 128 that is four short iterations, each
// exposing a global-load latency and a barrier, per 128 rows -- 600 workgroups, 2.3 per CU, every one of them staging the same
// 64 KB of weights again: 55 us per launch where the rows are 20 us of HBM traffic.
//
// Here ONE 8-wave workgroup per CU stages the weight once -- split into the two binary16 planes of 2^8 W, stored as MFMA B
// fragments ([16-channel step][32-column tile][plane][lane slot]: a fragment read is one conflict-free ds_read_b128) --
// and then every wave walks 32-row tiles of the flat row index (tile = wave, wave + #waves, ...): a tile's rows are loaded one
// tile AHEAD into registers (bounds-checked buffer loads: rows past the end read 0), split into the A operand's planes, and
// multiplied against the resident fragments: no barrier after the staging, no LDS traffic for the activations.  The epilogue is
// convgemm.h's (bias, activation, residual, LayerNorm, row mask, the data gradient's power-of-two scale).
//
// KS = c_in / 16 (5: the mel Linear's data gradient, 8: the 128-channel layers), up to 128 output channels (NT = 4 column tiles).
// AMP: `precision = 16` -- both operands rounded to binary16, one product.
#pragma once
#include "convgemm.h"

namespace esmi {

#if ESMI_CHAIN_SPLIT
constexpr int kPwWaves = 8;
template <int KS>
__host__ __device__ constexpr int pwgemm_lds_bytes() { return KS * 4 * 2 * 64 * 16; }

template <int KS, int NT, bool AMP>   // NT = 2: a wave's item is (32 rows, 64 columns); NT = 4: (32 rows, all columns) -- LayerNorm / row-dot epilogues
__global__ __launch_bounds__(64 * kPwWaves) void pwgemm_kernel(const ConvGemmP p, int n_items) {
    constexpr int NTW = 4, NH = NTW / NT;             // column tiles of the staged weight; items per row tile
    ESMI_DYN_LDS(lds);
    u32x4* wl = reinterpret_cast<u32x4*>(lds);        // [st][nt][plane][lane]
    const int tid = (int)threadIdx.x, lane = lane_id(), w = uniform_i(wave_id()), i = lane & 31, h = lane >> 5;
#ifdef ESMI_GEMM_TRACE   // development: shader-clock stamps of one mid-grid workgroup (tools/trace_pwgemm.py): start, staged, then 4 per tile
    const bool tr_on = g_gemm_trace_dev && blockIdx.x == gridDim.x / 2 && lane == 0;
    int tr_n = 0;
#define ESMI_PT() do { if (tr_on && tr_n < 64) g_gemm_trace_dev[w * 64 + tr_n] = (long long)__builtin_amdgcn_s_memtime(); ++tr_n; } while (0)
#else
#define ESMI_PT() do {} while (0)
#endif
    ESMI_PT();
    // ---- the rows (the first tile's loads are issued in front of the weight staging: in flight under it)
    const long n_rows = (long)p.B * p.n_out;
    const BufRsrc r_a = make_rsrc(p.A, ((n_rows - 1) * p.lda + p.a_coff + p.c_in) * 4L);
    constexpr unsigned kBig = 0x7FFFFFFFu;
    const float in_s = conv_in_scale(p);
    const int n_waves = (int)gridDim.x * kPwWaves;
    f32x4 nxt[KS][2];
    auto load_tile = [&](int item) __attribute__((always_inline)) {   // lane (i, h): row 32 (item / NH) + i, channels 16 st + 8 h + 0..7
        const long r = (long)(item / NH) * 32 + i;
        const unsigned off = r < n_rows ? (unsigned)((r * p.lda + p.a_coff + 8 * h) * 4L) : kBig;
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            nxt[st][0] = buf_ld4(r_a, off + 64u * st);
            nxt[st][1] = buf_ld4(r_a, off + 64u * st + 16u);
        }
    };
    load_tile((int)blockIdx.x * kPwWaves + w);
    ESMI_PT();   // first rows issued
    // ---- the weight, once.  A thread's item is 32 bytes of a weight row -- (row n, 16-channel step st, half hh): 16 consecutive threads
    // read one 512-byte row, i.e. a load instruction touches 16 cache lines (the TA takes about a cycle per distinct line: with a
    // lane per ROW, 64 lines per instruction, the staging was 9,000 cycles of a 25,000-cycle kernel in a trace) -- and lands in
    // B fragment (st, n / 32) at lane slot 32 hh + ((n + st + 8 hh) mod 32): rotated by (st, hh) so that the 16 threads of a row,
    // whose fragments are whole KiB apart, write 16 different 16-byte bank groups.  The reader applies the same rotation (a
    // permutation inside each half wave: its ds_read_b128 stay conflict-free).  All loads first, then the splits and the writes.
    {
        constexpr int NROW = NTW * 32, NQ = NROW * 2 * KS, NI = (NQ + 64 * kPwWaves - 1) / (64 * kPwWaves);
        f32x4 x0[NI], x1[NI];
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int q = tid + 64 * kPwWaves * u, n = q / (2 * KS), pc = q % (2 * KS);
            x0[u] = zero4(); x1[u] = zero4();
            if (q < NQ && n < p.c_out) {
                x0[u] = ld4(p.W + (long)n * p.c_in + 8 * pc);
                x1[u] = ld4(p.W + (long)n * p.c_in + 8 * pc + 4);
            }
        }
        ESMI_PT();   // weight loads issued
#ifdef ESMI_GEMM_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ESMI_PT();   // everything arrived
#endif
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int q = tid + 64 * kPwWaves * u, n = q / (2 * KS), pc = q % (2 * KS), st = pc >> 1, hh = pc & 1;
            if (q >= NQ) continue;
            const f32x4 y0 = x0[u] * kF16WScale, y1 = x1[u] * kF16WScale;
            u32x4 h1, h2;
            unsigned a, b;
            split_f16_pair_rn(y0[0], y0[1], a, b); h1[0] = a; h2[0] = b;   // weights: nearest-rounded pieces, as the pack-time splitters
            split_f16_pair_rn(y0[2], y0[3], a, b); h1[1] = a; h2[1] = b;
            split_f16_pair_rn(y1[0], y1[1], a, b); h1[2] = a; h2[2] = b;
            split_f16_pair_rn(y1[2], y1[3], a, b); h1[3] = a; h2[3] = b;
            const int f = st * NTW + (n >> 5), slot = 32 * hh + ((n + st + 8 * hh) & 31);
            wl[(f * 2 + 0) * 64 + slot] = h1;
            wl[(f * 2 + 1) * 64 + slot] = h2;
        }
    }
    ESMI_PT();   // weight written
    __syncthreads();
    ESMI_PT();
    const int rot = opaque_i(i + 8 * h);               // this lane's slot in step st: 32 h + ((i + 8 h + st) & 31)
    for (int item = (int)blockIdx.x * kPwWaves + w; item < n_items; item += n_waves) {
        const int cn = (item % NH) * NT;                // first column tile of the item
        f16x2p a[KS];
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            const f32x4 lo = nxt[st][0] * in_s, hi = nxt[st][1] * in_s;   // (in_s: the data gradient's power of two, else 1 -- exact)
            if constexpr (AMP) a[st].h1 = round_f16x8(lo, hi);
            else a[st] = split_f16x2(lo, hi);
        }
        sched_fence();
        ESMI_PT();   // rows arrived + split
        load_tile(item + n_waves);                     // the next item's rows: in flight under this item's products and stores (past the end: zeros)
        sched_fence();
        ESMI_PT();   // next loads issued
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = zero16();
        // B fragments in groups of two column tiles, one group ahead of the products that use them
        constexpr int NG = KS * (NT / 2);
        u32x4 bf[2][2][2];                             // [group parity][tile of the pair][plane]
        auto ld_b = [&](int g) __attribute__((always_inline)) {
            const int st = g / (NT / 2), np = g % (NT / 2);
            const u32x4* bl = wl + 32 * h + ((rot + st) & 31);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bf[g & 1][q][0] = bl[((st * NTW + cn + 2 * np + q) * 2 + 0) * 64];
                if constexpr (!AMP) bf[g & 1][q][1] = bl[((st * NTW + cn + 2 * np + q) * 2 + 1) * 64];
            }
        };
        ld_b(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) ld_b(g + 1);
            sched_fence();
            const int st = g / (NT / 2), np = g % (NT / 2);
            if constexpr (AMP) {
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[2 * np + q] = mfma32_f16(a[st].h1, bf[g & 1][q][0], acc[2 * np + q]);
            } else {
                // the three products of a tile are a dependent chain on its accumulator: run the two chains side by side
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x16& c = acc[2 * np + q];
                        c = pr == 0 ? mfma32_f16(a[st].h2, bf[g & 1][q][0], c) : pr == 1 ? mfma32_f16(a[st].h1, bf[g & 1][q][1], c) : mfma32_f16(a[st].h1, bf[g & 1][q][0], c);
                    }
            }
            sched_fence();
        }
        ESMI_PT();   // products issued
        convgemm_epilogue<NT>(acc, p, 0, (item / NH) * 32, 32 * cn, lane, 1, (int)n_rows);
        ESMI_PT();   // epilogue issued
    }
}
#undef ESMI_PT
#endif

}  // namespace esmi
