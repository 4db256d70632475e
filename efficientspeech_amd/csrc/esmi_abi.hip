// esmi C-ABI: the host-side launch sequences behind include/esmi.h.
// Built by `hipcc --offload-arch=gfx950` into libesmi.so (product) and, unchanged, by the host
// clang++ with -DESMI_WAVESIM into libesmi_sim.so (CPU wave simulator used only by tests).
#include "../../include/esmi.h"

#include <cmath>
#include <cstring>

#include "attention.h"
#include "convgemm.h"
#include "hifigan_resblock.h"
#include "train_ops.h"
#include "enc_attn_ffn.h"
#include "enc_fuse_va.h"
#include "enc_merge_qkv.h"
#include "esmi_dev.h"
#include "mel_decoder.h"
#include "mel_decoder_rows.h"
#include "small_kernels.h"

#ifndef ESMI_GEMM_LDS_MIN_ROWS   // rows (B * n_out) from which the per-op plan's GEMMs take the LDS-staged kernel
#ifdef ESMI_WAVESIM
#define ESMI_GEMM_LDS_MIN_ROWS 1   // the simulator tests are small: run them through it too
#else
#define ESMI_GEMM_LDS_MIN_ROWS 2048
#endif
#endif

using namespace esmi;

namespace {

inline hipStream_t S(esmi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? ESMI_OK : (int)e;
}
inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: remember per (instantiation, device ordinal) that
// it was made (once: keeps the call out of hipGraph captures).  `done` is one static table per call site.
constexpr int kMaxDevices = 64;
struct AttrOnce { bool done[kMaxDevices] = {}; };
inline int raise_lds_limit(const void* fn, AttrOnce& once) {
#ifndef ESMI_WAVESIM
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= kMaxDevices) return ESMI_ERR_UNSUPPORTED;
    if (!once.done[dev]) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        once.done[dev] = true;
    }
#endif
    return ESMI_OK;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

ConvGemmP conv_defaults() {
    ConvGemmP p;
    memset(&p, 0, sizeof p);
    p.k = 1; p.stride = 1; p.pad = 0; p.mode = MODE_CONV;
    return p;
}

// full_row: the epilogue needs a whole output row inside one wave (LayerNorm / row-dot)
int launch_convgemm(ConvGemmP p, hipStream_t st) {
    if ((p.c_in & 7) || p.c_in <= 0 || p.c_out <= 0 || p.n_out <= 0 || p.B <= 0) return ESMI_ERR_ARG;
    if (!p.W || !aligned16(p.W)) return ESMI_ERR_ARG;
    if (p.ids) {
        if (!p.table || (p.ld_table & 3) || !aligned16(p.table)) return ESMI_ERR_ARG;
    } else if (!p.A || (p.lda & 3) || (p.a_coff & 3) || !aligned16(p.A)) return ESMI_ERR_ARG;
    const bool full_row = p.ln_g || p.dot_out;
    if (p.c_out == 1 && p.mode == MODE_CONV && p.stride == 1 && !p.ids && !full_row && !p.res && !p.rowmask && p.out) {
        const long n = (long)p.B * p.n_out;
        ESMI_LAUNCH(conv_to1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
        return launch_status();
    }
    int nt;
    if (full_row) {
        nt = (p.c_out + 31) / 32;
        if (nt == 3) nt = 4;
        if (nt > 4 && nt <= 8) nt = 8;
        if (nt > 8) return ESMI_ERR_UNSUPPORTED;
        if (p.ln_g && p.c_out != 32 * nt) return ESMI_ERR_UNSUPPORTED;  // LN width must be 32/64/128/256
    } else {
        nt = p.c_out > 64 ? 4 : (p.c_out > 32 ? 2 : 1);
    }
#if ESMI_CHAIN_SPLIT
    // large plain convolutions / Linears: weight tile staged through LDS once per 128 positions (convgemm.h)
    if (p.mode == MODE_CONV && p.stride == 1 && !p.ids && (p.c_in & 31) == 0 && p.c_out > 64 && (long)p.B * p.n_out >= ESMI_GEMM_LDS_MIN_ROWS) {
        const int nl = full_row ? (nt <= 4 ? 4 : 8) : ((p.c_out & 255) == 0 ? 8 : 4);
        constexpr int kRows = 32 * ESMI_GEMM_LDS_WAVES;
        dim3 g2((unsigned)(p.B * ((p.n_out + kRows - 1) / kRows)), full_row ? 1 : (p.c_out + 32 * nl - 1) / (32 * nl));
        if (nl == 4) {
            ESMI_LAUNCH((convgemm_lds_kernel<4>), g2, dim3(64 * ESMI_GEMM_LDS_WAVES), convgemm_lds_bytes<4>(), st, p);
        } else {
            static AttrOnce once;
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(convgemm_lds_kernel<8>), once)) return rc;
            ESMI_LAUNCH((convgemm_lds_kernel<8>), g2, dim3(64 * ESMI_GEMM_LDS_WAVES), convgemm_lds_bytes<8>(), st, p);
        }
        return launch_status();
    }
#endif
    const int tiles = p.B * convgemm_tiles_per_phase(p) * convgemm_row_stride(p);
    dim3 grid((tiles + 3) / 4, full_row ? 1 : (p.c_out + 32 * nt - 1) / (32 * nt));
    dim3 block(256);
    switch (nt) {
        case 1: ESMI_LAUNCH((convgemm_kernel<1>), grid, block, 0, st, p); break;
        case 2: ESMI_LAUNCH((convgemm_kernel<2>), grid, block, 0, st, p); break;
        case 4: ESMI_LAUNCH((convgemm_kernel<4>), grid, block, 0, st, p); break;
        case 8: ESMI_LAUNCH((convgemm_kernel<8>), grid, block, 0, st, p); break;
        default: return ESMI_ERR_UNSUPPORTED;
    }
    return launch_status();
}

#ifndef ESMI_ATTN_LDS_MIN_HEADS
#ifdef ESMI_WAVESIM
#define ESMI_ATTN_LDS_MIN_HEADS 1      // (simulator: always take the LDS kernel where it applies, so the tests reach it)
#else
#define ESMI_ATTN_LDS_MIN_HEADS 128    // enough (utterance, head) workgroups to occupy the chip at one per CU
#endif
#endif
int launch_attn(const AttnP& p, hipStream_t st) {
    if ((p.C & 31) || p.N <= 0) return ESMI_ERR_ARG;   // channel groups of 32 (4 k-steps fetched together)
    const int nkt = (p.N + 31) / 32;
    const int tiles = p.B * p.h * nkt;
    dim3 grid((tiles + 3) / 4), block(256);
    if (nkt > 8) {   // N > 256: key-chunked two-sweep kernel (no sequence limit, as the reference)
        switch (p.C / 32) {
            case 1: ESMI_LAUNCH((attn_long_kernel<1>), grid, block, 0, st, p); break;
            case 2: ESMI_LAUNCH((attn_long_kernel<2>), grid, block, 0, st, p); break;
            case 4: ESMI_LAUNCH((attn_long_kernel<4>), grid, block, 0, st, p); break;
            case 8: ESMI_LAUNCH((attn_long_kernel<8>), grid, block, 0, st, p); break;
            default: return ESMI_ERR_UNSUPPORTED;   // widths of the three published sizes: 32 .. 256
        }
        return launch_status();
    }
#if ESMI_CHAIN_SPLIT
    // heads with several query tiles: K and V staged once per (utterance, head) in LDS instead of once per tile from L2
    if (nkt >= 3 && (p.C <= 128 || p.C % 128 == 0) && attn_lds_bytes(p.N, p.C) <= 150 * 1024 && (long)p.B * p.h >= ESMI_ATTN_LDS_MIN_HEADS) {
        const size_t lds = attn_lds_bytes(p.N, p.C);
        dim3 g2((unsigned)(p.B * p.h));
        if (nkt <= 4) {
            static AttrOnce once;
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(attn_lds_kernel<4>), once)) return rc;
            ESMI_LAUNCH((attn_lds_kernel<4>), g2, dim3(256), lds, st, p);
        } else {
            static AttrOnce once;
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(attn_lds_kernel<8>), once)) return rc;
            ESMI_LAUNCH((attn_lds_kernel<8>), g2, dim3(512), lds, st, p);
        }
        return launch_status();
    }
#endif
    if (nkt == 1) ESMI_LAUNCH((attn_kernel<1>), grid, block, 0, st, p);
    else if (nkt == 2) ESMI_LAUNCH((attn_kernel<2>), grid, block, 0, st, p);
    else if (nkt <= 4) ESMI_LAUNCH((attn_kernel<4>), grid, block, 0, st, p);
    else ESMI_LAUNCH((attn_kernel<8>), grid, block, 0, st, p);
    return launch_status();
}

inline int conv_out_len(int n, int k, int stride, int pad) { return (n + 2 * pad - k) / stride + 1; }


// E1: merge conv + 1x1 + qkv in one launch.  Returns ESMI_ERR_UNSUPPORTED when no instantiation fits.
int launch_enc_merge_qkv(const EncMergeP& p, int c_in, int c_out, hipStream_t st) {
    const int nci = c_in / 32, nc = c_out / 32;
    if ((c_in & 31) || (c_out & 31)) return ESMI_ERR_UNSUPPORTED;
    dim3 grid(p.B * p.tiles_per_b), block(64);
    const int lds = enc_merge_lds_floats(c_in, c_out, p.k, p.stride) * (int)sizeof(float);
#define ESMI_E1(NCI, NC, KT, ST) \
    if (nci == NCI && nc == NC && p.k == KT && p.stride == ST) { ESMI_LAUNCH((enc_merge_qkv_kernel<NCI, NC, KT, ST>), grid, block, lds, st, p); return launch_status(); }
    // (Cin/32, C/32, kernel, stride) of the three published sizes: tiny, small, base (block 1 of base is not fused)
    ESMI_E1(4, 1, 3, 1) ESMI_E1(1, 2, 1, 2) ESMI_E1(4, 2, 3, 1) ESMI_E1(2, 4, 1, 2) ESMI_E1(4, 4, 5, 1)
#undef ESMI_E1
    return ESMI_ERR_UNSUPPORTED;
}

bool enc_attn_ffn_supported(int C, int N, int expansion) {
    if ((C & 31) || N > 256 || N < 1) return false;
    // sequences of more than 128 positions: the chain kernel runs one latency chain per 32 rows against up to 256 keys; the same
    // ops as launches (LDS-staged attention + LDS-staged GEMMs) are faster there (small ES T = 256: 2.53 vs 2.62 ms/step; base ES
    // block 0: 1.33 vs 2.00 ms) -- `tools/debug_plan_base.py` measures the plans
    if (N > 128) return false;
    const int nc = C / 32;
    // base ES block 0 (C = 128, expansion 2) at N = 256: the chain kernel runs one latency chain per 32 rows against 256 keys
    // (2.00 ms at B = 512); the same ops as LDS-staged GEMM launches take 1.33 ms, so that shape goes per-op
    return (expansion == 1 && (nc == 1 || nc == 2 || nc == 4)) || (expansion == 2 && nc == 4 && N <= 128);
}

// Whole encoder block (merge conv + qkv + attention + MixFFN) in one launch: sequences one workgroup covers, shapes
// whose q/k/v tile fits in LDS.  Returns ESMI_ERR_UNSUPPORTED otherwise (-> enc_merge_qkv + enc_attn_ffn launches).
int launch_enc_block(const EncAttnFfnP& p, int expansion, int c_in, int plan, hipStream_t st) {
    if ((p.C & 31) || (c_in & 31) || p.N > 128) return ESMI_ERR_UNSUPPORTED;
    const int nc = p.C / 32, nci = c_in / 32, nkt = p.N <= 64 ? 2 : 4;
    if (p.h == 2 && nc == 2 && expansion == 1 && (plan & ESMI_FUSE_SPLIT2)) {   // two waves per row tile when rows are scarce
        int nw, wgs, useful, halo;
        enc_attn_ffn_split_plan(p.N, &nw, &wgs, &useful, &halo);
        if ((long)p.B * wgs * 2 * nw <= 1024) {
            const int lds = enc_block_split_lds_floats(p.C, p.h, expansion, c_in, p.m.k, p.m.stride, nw) * (int)sizeof(float);
            if (halo != 0 || lds > 150 * 1024 || !(nci == 1 && p.m.k == 1 && p.m.stride == 2)) return ESMI_ERR_UNSUPPORTED;
            EncAttnFfnP q = p;
            q.wgs_per_b = 1; q.useful = useful; q.halo = 0;
            dim3 grid(p.B), block(128 * nw);
            static AttrOnce once;
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(enc_attn_ffn_split_kernel<2, 2, 1, 1, 1, 2>), once)) return rc;
            ESMI_LAUNCH((enc_attn_ffn_split_kernel<2, 2, 1, 1, 1, 2>), grid, block, lds, st, q);   // N <= 64 here: NKT = 2
            return launch_status();
        }
    }
    int nw, wgs, useful, halo;
    enc_attn_ffn_plan(p.N, p.C * expansion + 4, &nw, &wgs, &useful, &halo);
    if (halo != 0) return ESMI_ERR_UNSUPPORTED;
    const int lds = enc_block_lds_floats(p.C, p.h, expansion, c_in, p.m.k, p.m.stride, nw) * (int)sizeof(float);
    if (lds > 150 * 1024) return ESMI_ERR_UNSUPPORTED;
    EncAttnFfnP q = p;
    q.wgs_per_b = 1; q.useful = useful; q.halo = 0;
    dim3 grid(p.B), block(64 * nw);
#define ESMI_EB(NKT, NC, E, NCI, KT, ST) \
    if (nkt == NKT && nc == NC && expansion == E && nci == NCI && p.m.k == KT && p.m.stride == ST) {                           \
        static AttrOnce once; /* per instantiation */                                                                          \
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(enc_attn_ffn_kernel<NKT, NC, E, NCI, KT, ST>), once)) return rc; \
        ESMI_LAUNCH((enc_attn_ffn_kernel<NKT, NC, E, NCI, KT, ST>), grid, block, lds, st, q);                                  \
        return launch_status();                                                                                                \
    }
    // tiny block 0 / small block 0 / tiny block 1 (when the split kernel does not apply)
    ESMI_EB(2, 1, 1, 4, 3, 1) ESMI_EB(4, 1, 1, 4, 3, 1) ESMI_EB(2, 2, 1, 4, 3, 1) ESMI_EB(4, 2, 1, 4, 3, 1)
    ESMI_EB(2, 2, 1, 1, 1, 2)
#undef ESMI_EB
    return ESMI_ERR_UNSUPPORTED;
}

// E2: attention + proj + LN1 + MixFFN + LN2 in one launch.
int launch_enc_attn_ffn(const EncAttnFfnP& p, int expansion, int plan, hipStream_t st) {
    if ((p.C & 31) || p.N > 256) return ESMI_ERR_UNSUPPORTED;
    const int nc = p.C / 32, nkt = p.N <= 64 ? 2 : (p.N <= 128 ? 4 : 8);
    int nw, wgs, useful, halo;
    EncAttnFfnP q = p;
    if (p.h == 2 && nc == 2 && expansion == 1 && (plan & ESMI_FUSE_SPLIT2)) {   // two waves per row tile when rows are scarce
        enc_attn_ffn_split_plan(p.N, &nw, &wgs, &useful, &halo);
        if ((long)p.B * wgs * 2 * nw <= 1024 && nkt <= 4) {
            q.wgs_per_b = wgs; q.useful = useful; q.halo = halo;
            dim3 grid(p.B * wgs), block(128 * nw);
            const int lds = enc_attn_ffn_split_lds_floats(p.C, p.h, expansion, nw) * (int)sizeof(float);
            if (nkt == 2) ESMI_LAUNCH((enc_attn_ffn_split_kernel<2, 2, 1>), grid, block, lds, st, q);
            else ESMI_LAUNCH((enc_attn_ffn_split_kernel<4, 2, 1>), grid, block, lds, st, q);
            return launch_status();
        }
    }
    enc_attn_ffn_plan(p.N, p.C * expansion + 4, &nw, &wgs, &useful, &halo);
    q.wgs_per_b = wgs; q.useful = useful; q.halo = halo;
    dim3 grid(p.B * wgs), block(64 * nw);
    const int lds = (32 * nw + 2) * (p.C * expansion + 4) * (int)sizeof(float);
#define ESMI_E2(NKT, NC, E) \
    if (nkt == NKT && nc == NC && expansion == E) {                                                                            \
        static AttrOnce once; /* per instantiation */                                                                          \
        if (lds > 48 * 1024)                                                                                                   \
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(enc_attn_ffn_kernel<NKT, NC, E>), once)) return rc;     \
        ESMI_LAUNCH((enc_attn_ffn_kernel<NKT, NC, E>), grid, block, lds, st, q);                                               \
        return launch_status();                                                                                                \
    }
#define ESMI_E2K(NC, E) ESMI_E2(2, NC, E) ESMI_E2(4, NC, E) ESMI_E2(8, NC, E)
    ESMI_E2K(1, 1) ESMI_E2K(2, 1) ESMI_E2K(4, 1) ESMI_E2K(4, 2)
#undef ESMI_E2K
#undef ESMI_E2
    return ESMI_ERR_UNSUPPORTED;
}

// does the fused Fuse + variance-adaptor chain kernel serve this call?  (needs the packed weights and one of its instantiations)
bool fuse_va_chain_ok(const esmi_fuse_weights* fw, int depth, int dim, int kernel, int n0, int T, const esmi_predictor_weights* pitch,
                      const esmi_predictor_weights* energy, const esmi_predictor_weights* duration, int plan) {
    bool chain = (plan & ESMI_FUSE_VARIANCE) && (dim == 32 || dim == 64) && (kernel == 3 || kernel == 5) && n0 == T && fw->fuse_wp &&
                 pitch->conv1_wp && pitch->conv2_wp && energy->conv1_wp && energy->conv2_wp && duration->conv1_wp &&
                 duration->conv2_wp;
    for (int i = 0; i < depth && chain; ++i) chain = fw->mlp_wp[i] && (i == 0 || fw->up_wp[i]);
    return chain;
}
// ... and can it also produce the mel decoder's first stage at phoneme rate?
bool fuse_va_head_ok(bool chain, int dim, const esmi_decoder_head* head) {
    return chain && head && head->proj_wp && dim == 32 && head->d4 == 128 && head->dx2 == 128;
}

struct EncWs {
    size_t t_merge, qkv, ctx, y1, m1, m2, pmask, total;
};
EncWs enc_ws(const esmi_encoder_block_shape* s) {
    const int n = conv_out_len(s->n_in, s->kernel, s->stride, s->kernel / 2);
    const size_t rows = (size_t)s->B * n;
    EncWs w;
    size_t o = 0;
    w.t_merge = o; o += align256(rows * s->c_in * 4);
    w.qkv = o; o += align256(rows * 3 * s->heads * s->c_out * 4);
    w.ctx = o; o += align256(rows * s->heads * s->c_out * 4);
    w.y1 = o; o += align256(rows * s->c_out * 4);
    w.m1 = o; o += align256(rows * s->c_out * s->expansion * 4);
    w.m2 = o; o += align256(rows * s->c_out * s->expansion * 4);
    w.pmask = o; o += align256(rows);
    w.total = o;
    return w;
}

}  // namespace

#ifdef ESMI_CHAIN_TRACE
__device__ long long* g_chain_trace_dev = nullptr;
extern "C" void esmi_dev_set_chain_trace(long long* ptr) {
    hipMemcpyToSymbol(HIP_SYMBOL(g_chain_trace_dev), &ptr, sizeof(ptr));
}
#endif
#ifdef ESMI_DEC_TRACE
long long* g_esmi_trace = nullptr;
extern "C" void esmi_dev_set_trace(long long* ptr) { g_esmi_trace = ptr; }
#endif

namespace {
// One ResBlock in one launch (hifigan_resblock.h) when its packed weights are there and (channels, kernel size) has an
// instantiation; `false` from resblock_fused_ok -> the caller runs the block conv by conv.
template <int C, int K>
int launch_resblock_ck(const ResblockP& p, hipStream_t st) {
    const size_t lds = rb_lds_bytes(C, p.R);
    const dim3 grid((unsigned)(p.B * p.tiles_per_b)), block(64 * kRbWaves);
    if constexpr (C <= 16) {   // narrow MFMA tiles (16 channels x 16 positions): LDS <= 32 KB, no limit to raise
        ESMI_LAUNCH((hifigan_resblock16_kernel<C, K>), grid, block, lds, st, p);
    } else {
        static AttrOnce once;
        if (lds > 48 * 1024)
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(hifigan_resblock_kernel<C, K>), once)) return rc;
        ESMI_LAUNCH((hifigan_resblock_kernel<C, K>), grid, block, lds, st, p);
    }
    return launch_status();
}
template <int C>
int launch_resblock_c(const ResblockP& p, hipStream_t st) {
    switch (p.k) {
        case 3: return launch_resblock_ck<C, 3>(p, st);
        case 7: return launch_resblock_ck<C, 7>(p, st);
        case 11: return launch_resblock_ck<C, 11>(p, st);
    }
    return ESMI_ERR_UNSUPPORTED;
}
int launch_resblock(const ResblockP& p, int c, hipStream_t st) {
    switch (c) {
        case 8: return launch_resblock_c<8>(p, st);
        case 16: return launch_resblock_c<16>(p, st);
        case 32: return launch_resblock_c<32>(p, st);
        case 64: return launch_resblock_c<64>(p, st);
    }
    return ESMI_ERR_UNSUPPORTED;
}
}  // namespace

extern "C" {

int esmi_version(void) { return ESMI_VERSION; }
const char* esmi_backend(void) {
#ifdef ESMI_WAVESIM
    return "wavesim";
#else
    return "hip:gfx950";
#endif
}

const char* esmi_build_config(void) {
#if ESMI_CHAIN_SPLIT
#define ESMI_CFG_ENC_ ",enc_gemm=split-f16x2"
#else
#define ESMI_CFG_ENC_ ",enc_gemm=fp32-mfma"
#endif
#if ESMI_DEC_SPLIT == 2
    return "dec_gemm=split-f16x2" ESMI_CFG_ENC_;
#else
    return "dec_gemm=fp32-mfma" ESMI_CFG_ENC_;
#endif
}

int esmi_pack_conv_weight_f32(const float* src, float* dst, int cout, int cin, int k, esmi_stream_t stream) {
    if (!src || !dst || cout <= 0 || cin <= 0 || k <= 0) return ESMI_ERR_ARG;
    const long n = (long)cout * cin * k;
    ESMI_LAUNCH(pack_conv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), src, dst, cout, cin, k, 0);
    return launch_status();
}
int esmi_pack_convT_weight_f32(const float* src, float* dst, int cin, int cout, int k, esmi_stream_t stream) {
    if (!src || !dst || cout <= 0 || cin <= 0 || k <= 0) return ESMI_ERR_ARG;
    const long n = (long)cout * cin * k;
    ESMI_LAUNCH(pack_conv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), src, dst, cout, cin, k, 1);
    return launch_status();
}

size_t esmi_pack_bfrag_floats(int n, int k, int taps) {
    if (n <= 0 || k <= 0 || taps <= 0 || (k & 7)) return 0;
    return (size_t)taps * k * 32 * ((n + 31) / 32);
}
int esmi_pack_bfrag_f32(const float* src, float* dst, int n, int k, int taps, esmi_stream_t stream) {
    if (!src || !dst || n <= 0 || k <= 0 || taps <= 0 || (k & 7)) return ESMI_ERR_ARG;
#if ESMI_CHAIN_SPLIT
    if (k & 31) return ESMI_ERR_UNSUPPORTED;   // the split-f16 packing works on groups of 32 channels (two 16-channel MFMA steps)
#endif
    const long tot = (long)esmi_pack_bfrag_floats(n, k, taps);
    ESMI_LAUNCH(pack_bfrag_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, S(stream), src, dst, n, k, (n + 31) / 32, taps);
    return launch_status();
}

size_t esmi_pack_resblock_bytes(int c, int k) {
    if ((c != 8 && c != 16 && c != 32 && c != 64) || (k != 3 && k != 7 && k != 11)) return 0;
    return rb_pack_dwords(c, k) * 4;
}
int esmi_pack_resblock_f16(const float* src, void* dst, int c, int k, esmi_stream_t stream) {
    if (!src || !dst) return ESMI_ERR_ARG;
    if (!esmi_pack_resblock_bytes(c, k)) return ESMI_ERR_UNSUPPORTED;
    if (c <= 16) {
        const long n16 = (long)rb_ksteps16(c, k) * 64;
        ESMI_LAUNCH(pack_resblock16_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, S(stream), src, static_cast<unsigned*>(dst), c, k);
        return launch_status();
    }
    const long n = (long)rb_mtiles(c) * rb_ksteps(c, k) * 64;
    ESMI_LAUNCH(pack_resblock_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), src, static_cast<unsigned*>(dst), c, k);
    return launch_status();
}

int esmi_compose_merge_f32(const float* merge_w, const float* merge1_w, int k, int cin, int cout, float* dst,
                           esmi_stream_t stream) {
    if (!merge_w || !merge1_w || !dst || k <= 0 || cin <= 0 || cout <= 0) return ESMI_ERR_ARG;
    const long n = (long)k * cin * cout;
    ESMI_LAUNCH(compose_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), merge_w, merge1_w, dst, k, cin, cout);
    return launch_status();
}

int esmi_pool_mask_u8(const uint8_t* mask, int B, int T, int pool, uint8_t* out, int n_out, esmi_stream_t stream) {
    if (!mask || !out || B <= 0 || T <= 0 || pool <= 0 || n_out <= 0) return ESMI_ERR_ARG;
    ESMI_LAUNCH(pool_mask_kernel, dim3((B * n_out + 255) / 256), dim3(256), 0, S(stream), mask, B, T, pool, out, n_out);
    return launch_status();
}

size_t esmi_encoder_block_workspace_bytes(const esmi_encoder_block_shape* s) { return s ? enc_ws(s).total : 0; }

int esmi_encoder_block_f32(const esmi_encoder_block_weights* w, const esmi_encoder_block_shape* s, const int32_t* ids,
                           const float* embed, const float* x_in, const uint8_t* mask, float* x_out, void* workspace,
                           size_t workspace_bytes, esmi_stream_t stream) {
    if (!w || !s || !x_out || !workspace) return ESMI_ERR_ARG;
    if (!ids && !x_in) return ESMI_ERR_ARG;
    const int plan = s->plan & ESMI_FUSE_ALL;
    const EncWs ws = enc_ws(s);
    if (workspace_bytes < ws.total) return ESMI_ERR_WORKSPACE;
    char* wsb = static_cast<char*>(workspace);
    float* t_merge = reinterpret_cast<float*>(wsb + ws.t_merge);
    float* qkv = reinterpret_cast<float*>(wsb + ws.qkv);
    float* ctx = reinterpret_cast<float*>(wsb + ws.ctx);
    float* y1 = reinterpret_cast<float*>(wsb + ws.y1);
    float* m1 = reinterpret_cast<float*>(wsb + ws.m1);
    float* m2 = reinterpret_cast<float*>(wsb + ws.m2);
    const int B = s->B, C = s->c_out, h = s->heads, E = s->c_out * s->expansion;
    const int n = conv_out_len(s->n_in, s->kernel, s->stride, s->kernel / 2);
    hipStream_t st = S(stream);
    int rc;
    ConvGemmP p = conv_defaults();
    bool fused1 = false;
    // With the fused second stage, x (the block input after the merge convs) lives in scratch and the final
    // result is written straight to x_out: tiles read their neighbours' x rows, so in-place is not possible.
    const bool packed = w->merge_cwp && w->qkv_wp && w->proj_wp && w->mlp1_wp && w->conv_wp && w->mlp2_wp;
    const bool fused2 = packed && (plan & ESMI_FUSE_ATTN_FFN) && enc_attn_ffn_supported(C, n, s->expansion);
    float* x_mid = fused2 ? y1 : x_out;
    EncMergeP m;
    memset(&m, 0, sizeof m);
    m.ids = ids; m.table = embed; m.vocab = s->vocab; m.x_in = ids ? nullptr : x_in;
    m.B = B; m.n_in = s->n_in; m.n_out = n; m.k = s->kernel; m.stride = s->stride; m.pad = s->kernel / 2; m.h = h;
    m.merge_w = w->merge_cwp; m.qkv_w = w->qkv_wp; m.x_out = x_mid; m.qkv = qkv;
    m.tiles_per_b = (n + 31) / 32;
    EncAttnFfnP f;
    memset(&f, 0, sizeof f);
    f.x = x_mid; f.qkv = qkv; f.B = B; f.N = n; f.C = C; f.h = h; f.scale = 1.0f / sqrtf((float)(C / h));
    f.proj_w = w->proj_wp; f.proj_b = w->proj_b; f.ln1_g = w->ln1_g; f.ln1_b = w->ln1_b;
    f.mlp1_w = w->mlp1_wp; f.mlp1_b = w->mlp1_b; f.conv_w = w->conv_wp; f.conv_b = w->conv_b;
    f.mlp2_w = w->mlp2_wp; f.mlp2_b = w->mlp2_b; f.ln2_g = w->ln2_g; f.ln2_b = w->ln2_b;
    f.mask = mask; f.out = x_out;
    f.mask_pool = s->mask_pool > 0 ? s->mask_pool : 1; f.mask_len = s->mask_pool > 0 ? s->mask_len : n;
    if (fused2 && (plan & ESMI_FUSE_MERGE_QKV) && (plan & ESMI_FUSE_BLOCK)) {   // the whole block in one launch
        f.m = m;
        f.x = nullptr; f.qkv = nullptr;
        rc = launch_enc_block(f, s->expansion, s->c_in, plan, st);
        if (rc != ESMI_ERR_UNSUPPORTED) return rc;
        f.x = x_mid; f.qkv = qkv;
    }
    if (packed && (plan & ESMI_FUSE_MERGE_QKV)) {   // E1: merge conv + 1x1 + qkv as one wave-chain kernel
        rc = launch_enc_merge_qkv(m, s->c_in, C, st);
        if (rc == ESMI_OK) fused1 = true;
        else if (rc != ESMI_ERR_UNSUPPORTED) return rc;
    }
    if (!fused1) {
        // merge conv k x k (dense, bias-free), networks.py:64-66
        p.B = B; p.n_in = s->n_in; p.c_in = s->c_in; p.n_out = n; p.c_out = s->c_in;
        p.k = s->kernel; p.stride = s->stride; p.pad = s->kernel / 2;
        if (ids) { p.ids = ids; p.table = embed; p.ld_table = s->c_in; p.vocab = s->vocab; }
        else { p.A = x_in; p.lda = s->c_in; }
        p.W = w->merge_w; p.out = t_merge; p.ldo = s->c_in;
        if ((rc = launch_convgemm(p, st))) return rc;
        // merge 1x1, networks.py:67
        p = conv_defaults();
        p.B = B; p.n_in = n; p.c_in = s->c_in; p.n_out = n; p.c_out = C;
        p.A = t_merge; p.lda = s->c_in; p.W = w->merge1_w; p.out = x_mid; p.ldo = C;
        if ((rc = launch_convgemm(p, st))) return rc;
        // qkv Linear (bias-free), blocks.py:44
        p = conv_defaults();
        p.B = B; p.n_in = n; p.c_in = C; p.n_out = n; p.c_out = 3 * h * C;
        p.A = x_mid; p.lda = C; p.W = w->qkv_w; p.out = qkv; p.ldo = 3 * h * C;
        if ((rc = launch_convgemm(p, st))) return rc;
    }
    if (fused2) {   // E2: attention + proj + LN1 + MixFFN + LN2 as one wave-chain kernel
        return launch_enc_attn_ffn(f, s->expansion, plan, st);
    }
    if (mask && s->mask_pool > 1) {   // the one-kernel-per-op plan takes a pooled (B, n) mask: blocks.py:51-57
        uint8_t* pm = reinterpret_cast<uint8_t*>(wsb + ws.pmask);
        if ((rc = esmi_pool_mask_u8(mask, B, s->mask_len, s->mask_pool, pm, n, stream))) return rc;
        mask = pm;
    }
    // softmax(q k^T scale) v, blocks.py:49-64
    AttnP a;
    a.qkv = qkv; a.B = B; a.N = n; a.C = C; a.h = h; a.ctx = ctx;
    a.scale = 1.0f / sqrtf((float)(C / h));
    if ((rc = launch_attn(a, st))) return rc;
    // proj + residual + LN1 + mask, blocks.py:65 + networks.py:73-75
    p = conv_defaults();
    p.B = B; p.n_in = n; p.c_in = h * C; p.n_out = n; p.c_out = C;
    p.A = ctx; p.lda = h * C; p.W = w->proj_w; p.bias = w->proj_b;
    p.res = x_out; p.ldr = C; p.ln_g = w->ln1_g; p.ln_b = w->ln1_b; p.rowmask = mask;
    p.out = y1; p.ldo = C;
    if ((rc = launch_convgemm(p, st))) return rc;
    // MixFFN, blocks.py:22-29
    p = conv_defaults();
    p.B = B; p.n_in = n; p.c_in = C; p.n_out = n; p.c_out = E;
    p.A = y1; p.lda = C; p.W = w->mlp1_w; p.bias = w->mlp1_b; p.out = m1; p.ldo = E;
    if ((rc = launch_convgemm(p, st))) return rc;
    p = conv_defaults();
    p.B = B; p.n_in = n; p.c_in = E; p.n_out = n; p.c_out = E; p.k = 3; p.pad = 1;
    p.A = m1; p.lda = E; p.W = w->conv_w; p.bias = w->conv_b; p.act = ACT_GELU; p.out = m2; p.ldo = E;
    if ((rc = launch_convgemm(p, st))) return rc;
    // mlp2 + residual + LN2 + mask, networks.py:80-83
    p = conv_defaults();
    p.B = B; p.n_in = n; p.c_in = E; p.n_out = n; p.c_out = C;
    p.A = m2; p.lda = E; p.W = w->mlp2_w; p.bias = w->mlp2_b;
    p.res = y1; p.ldr = C; p.ln_g = w->ln2_g; p.ln_b = w->ln2_b; p.rowmask = mask;
    p.out = x_out; p.ldo = C;
    return launch_convgemm(p, st);
}

size_t esmi_fuse_workspace_bytes(int B, int T, int dim, int depth) {
    return align256((size_t)B * T * dim * depth * 4) + align256((size_t)B * T * dim * 4);
}

int esmi_fuse_f32(const esmi_fuse_weights* w, int depth, int dim, int kernel, int B, int T, const float* const* feats,
                  const int* n_i, const uint8_t* mask, float* out, int ld_out, void* workspace, size_t workspace_bytes,
                  esmi_stream_t stream) {
    if (!w || !feats || !n_i || !out || !workspace || depth < 1 || depth > ESMI_MAX_DEPTH) return ESMI_ERR_ARG;
    if (workspace_bytes < esmi_fuse_workspace_bytes(B, T, dim, depth)) return ESMI_ERR_WORKSPACE;
    float* cat = static_cast<float*>(workspace);
    float* tmp = reinterpret_cast<float*>(static_cast<char*>(workspace) + align256((size_t)B * T * dim * depth * 4));
    hipStream_t st = S(stream);
    int rc;
    for (int i = 0; i < depth; ++i) {
        const int ci = dim << i, s = 1 << i;
        ConvGemmP p = conv_defaults();  // Linear(dim*2^i, dim), networks.py:197
        p.B = B; p.n_in = n_i[i]; p.c_in = ci; p.n_out = n_i[i]; p.c_out = dim;
        p.A = feats[i]; p.lda = ci; p.W = w->mlp_w[i]; p.bias = w->mlp_b[i];
        if (i == 0) {
            if (n_i[0] != T) return ESMI_ERR_ARG;
            p.out = cat; p.ldo = dim * depth; p.o_coff = 0;
            if ((rc = launch_convgemm(p, st))) return rc;
        } else {
            p.out = tmp; p.ldo = dim;
            if ((rc = launch_convgemm(p, st))) return rc;
            if ((n_i[i] - 1) * s + kernel < T) return ESMI_ERR_UNSUPPORTED;  // torch.cat would raise in the reference
            p = conv_defaults();  // ConvTranspose1d(dim, dim, k, stride 2^i) cropped to T, networks.py:199-206
            p.mode = MODE_CONVT; p.k = kernel; p.stride = s;
            p.B = B; p.n_in = n_i[i]; p.c_in = dim; p.n_out = T; p.c_out = dim;
            p.A = tmp; p.lda = dim; p.W = w->up_w[i]; p.bias = w->up_b[i];
            p.out = cat; p.ldo = dim * depth; p.o_coff = i * dim;
            if ((rc = launch_convgemm(p, st))) return rc;
        }
    }
    ConvGemmP p = conv_defaults();  // Linear(depth*dim, dim) + masked_fill, networks.py:215-217
    p.B = B; p.n_in = T; p.c_in = dim * depth; p.n_out = T; p.c_out = dim;
    p.A = cat; p.lda = dim * depth; p.W = w->fuse_w; p.bias = w->fuse_b; p.rowmask = mask;
    p.out = out; p.ldo = ld_out;
    return launch_convgemm(p, st);
}

size_t esmi_variance_adaptor_workspace_bytes(int B, int T, int dim) { return align256((size_t)B * T * dim * 4); }

int esmi_variance_adaptor_f32(const esmi_predictor_weights* pitch, const esmi_predictor_weights* energy,
                              const esmi_predictor_weights* duration, int dim, int B, int T, const uint8_t* mask,
                              const float* pitch_target, const float* energy_target, const int32_t* duration_target,
                              float* feat, float* pitch_pred, float* energy_pred, float* duration_pred,
                              int32_t* pitch_idx, int32_t* energy_idx, int32_t* dur, void* workspace,
                              size_t workspace_bytes, esmi_stream_t stream) {
    if (!pitch || !energy || !duration || !feat || !pitch_pred || !energy_pred || !duration_pred || !dur || !workspace)
        return ESMI_ERR_ARG;
    if (workspace_bytes < esmi_variance_adaptor_workspace_bytes(B, T, dim)) return ESMI_ERR_WORKSPACE;
    float* t1 = static_cast<float*>(workspace);
    hipStream_t st = S(stream);
    const esmi_predictor_weights* pw[3] = {pitch, energy, duration};
    float* preds[3] = {pitch_pred, energy_pred, duration_pred};
    int rc;
    for (int q = 0; q < 3; ++q) {
        // conv1 + ReLU -> LN1 -> ReLU, networks.py:152-155
        ConvGemmP p = conv_defaults();
        p.B = B; p.n_in = T; p.c_in = dim; p.n_out = T; p.c_out = dim; p.k = 3; p.pad = 1;
        p.A = feat; p.lda = 4 * dim; p.a_coff = 0; p.W = pw[q]->conv1_w; p.bias = pw[q]->conv1_b; p.act = ACT_RELU;
        p.ln_g = pw[q]->ln1_g; p.ln_b = pw[q]->ln1_b; p.post_relu = 1; p.out = t1; p.ldo = dim;
        if ((rc = launch_convgemm(p, st))) return rc;
        // conv2 + ReLU; pred = Linear(dim,1) on the PRE-norm2 tensor (:157-160); duration: ReLU + features = LN2
        p = conv_defaults();
        p.B = B; p.n_in = T; p.c_in = dim; p.n_out = T; p.c_out = dim; p.k = 3; p.pad = 1;
        p.A = t1; p.lda = dim; p.W = pw[q]->conv2_w; p.bias = pw[q]->conv2_b; p.act = ACT_RELU;
        p.dot_w = pw[q]->lin_w; p.dot_b = pw[q]->lin_b; p.dot_out = preds[q]; p.dot_relu = q == 2;
        if (q == 2) {
            p.ln_g = pw[q]->ln2_g; p.ln_b = pw[q]->ln2_b; p.rowmask = mask;   // :366-368
            p.out = feat; p.ldo = 4 * dim; p.o_coff = 3 * dim;
        }
        if ((rc = launch_convgemm(p, st))) return rc;
    }
    VaTailP v;
    v.rows = B * T; v.T = T; v.dim = dim; v.mask = mask;
    v.pitch_pred = pitch_pred; v.energy_pred = energy_pred; v.dur_pred = duration_pred;
    v.pitch_t = pitch_target; v.energy_t = energy_target; v.dur_t = duration_target;
    v.pbins = pitch->bins; v.ebins = energy->bins; v.pemb = pitch->emb; v.eemb = energy->emb;
    v.feat = feat; v.pitch_idx = pitch_idx; v.energy_idx = energy_idx; v.dur = dur;
    if (!v.pbins || !v.ebins || !v.pemb || !v.eemb) return ESMI_ERR_ARG;
    const long n = (long)B * T * dim;
    ESMI_LAUNCH(va_tail_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, v);
    return launch_status();
}

size_t esmi_fuse_variance_adaptor_workspace_bytes(int B, int T, int dim, int depth) {
    return esmi_fuse_workspace_bytes(B, T, dim, depth) + esmi_variance_adaptor_workspace_bytes(B, T, dim);
}

int esmi_fuse_variance_adaptor_f32(const esmi_fuse_weights* fw, int depth, int dim, int kernel, int B, int T,
                                   const float* const* feats, const int* n_i, const esmi_predictor_weights* pitch,
                                   const esmi_predictor_weights* energy, const esmi_predictor_weights* duration,
                                   const uint8_t* mask, const float* pitch_target, const float* energy_target,
                                   const int32_t* duration_target, float* feat, float* pitch_pred, float* energy_pred,
                                   float* duration_pred, int32_t* pitch_idx, int32_t* energy_idx, int32_t* dur,
                                   int32_t* cum, int32_t* mel_len, const esmi_decoder_head* head, float* h0, int plan,
                                   void* workspace, size_t workspace_bytes, esmi_stream_t stream) {
    if ((cum == nullptr) != (mel_len == nullptr)) return ESMI_ERR_ARG;
    if (h0 && (!head || !head->proj_wp || !head->proj_b || !head->ln_g || !head->ln_b)) return ESMI_ERR_ARG;
    if (!fw || !feats || !n_i || !pitch || !energy || !duration || !feat || !pitch_pred || !energy_pred ||
        !duration_pred || !dur || depth < 1 || depth > ESMI_MAX_DEPTH)
        return ESMI_ERR_ARG;
    const bool chain = fuse_va_chain_ok(fw, depth, dim, kernel, n_i[0], T, pitch, energy, duration, plan);
    if (h0 && !fuse_va_head_ok(chain, dim, head)) return ESMI_ERR_UNSUPPORTED;
    for (int i = 1; i < depth && chain; ++i)
        if ((n_i[i] - 1) * (1 << i) + kernel < T) return ESMI_ERR_UNSUPPORTED;   // torch.cat would raise in the reference
    if (chain) {
        FuseVaP p;
        memset(&p, 0, sizeof p);
        p.B = B; p.T = T; p.depth = depth; p.kernel = kernel;
        for (int i = 0; i < depth; ++i) {
            p.feats[i] = feats[i]; p.n_i[i] = n_i[i];
            p.mlp_w[i] = fw->mlp_wp[i]; p.mlp_b[i] = fw->mlp_b[i]; p.up_w[i] = fw->up_wp[i]; p.up_b[i] = fw->up_b[i];
        }
        p.fuse_w = fw->fuse_wp; p.fuse_b = fw->fuse_b;
        const esmi_predictor_weights* pw[3] = {pitch, energy, duration};
        for (int q = 0; q < 3; ++q) {
            PredW& d = p.pred[q];
            d.conv1_w = pw[q]->conv1_wp; d.conv1_b = pw[q]->conv1_b; d.ln1_g = pw[q]->ln1_g; d.ln1_b = pw[q]->ln1_b;
            d.conv2_w = pw[q]->conv2_wp; d.conv2_b = pw[q]->conv2_b; d.ln2_g = pw[q]->ln2_g; d.ln2_b = pw[q]->ln2_b;
            d.lin_w = pw[q]->lin_w; d.lin_b = pw[q]->lin_b; d.bins = pw[q]->bins; d.emb = pw[q]->emb;
        }
        if (!pitch->bins || !pitch->emb || !energy->bins || !energy->emb) return ESMI_ERR_ARG;
        p.mask = mask; p.pitch_t = pitch_target; p.energy_t = energy_target; p.dur_t = duration_target;
        p.feat = feat; p.preds[0] = pitch_pred; p.preds[1] = energy_pred; p.preds[2] = duration_pred;
        p.pitch_idx = pitch_idx; p.energy_idx = energy_idx; p.dur = dur;
        int nw;
        fuse_va_plan(T, dim, depth, &nw, &p.wgs_per_b, &p.useful, &p.halo);
        const bool scan_fused = cum && p.halo == 0;   // one workgroup sees every duration of its utterance
        p.cum = scan_fused ? cum : nullptr; p.mel_len = scan_fused ? mel_len : nullptr;
        dim3 grid(B * p.wgs_per_b), block(64 * nw);
        if (h0) { p.head_w = head->proj_wp; p.head_b = head->proj_b; p.head_g = head->ln_g; p.head_beta = head->ln_b; p.h0 = h0; }
        const int lds = fuse_va_lds_floats(dim, depth, nw, h0 != nullptr) * (int)sizeof(float);
        static AttrOnce once[4];
        const void* fns[4] = {reinterpret_cast<const void*>(enc_fuse_va_kernel<1, 3>), reinterpret_cast<const void*>(enc_fuse_va_kernel<2, 3>),
                              reinterpret_cast<const void*>(enc_fuse_va_kernel<1, 5>), reinterpret_cast<const void*>(enc_fuse_va_kernel<2, 5>)};
        for (int q = 0; q < 4; ++q)
            if (int rc = raise_lds_limit(fns[q], once[q])) return rc;
        if (dim == 32 && kernel == 3) ESMI_LAUNCH((enc_fuse_va_kernel<1, 3>), grid, block, lds, S(stream), p);
        else if (dim == 64 && kernel == 3) ESMI_LAUNCH((enc_fuse_va_kernel<2, 3>), grid, block, lds, S(stream), p);
        else if (dim == 32) ESMI_LAUNCH((enc_fuse_va_kernel<1, 5>), grid, block, lds, S(stream), p);
        else ESMI_LAUNCH((enc_fuse_va_kernel<2, 5>), grid, block, lds, S(stream), p);
        if (cum && !scan_fused) ESMI_LAUNCH(length_regulate_kernel, dim3(B), dim3(64), 0, S(stream), dur, T, cum, mel_len, (int*)nullptr);
        return launch_status();
    }
    if (!workspace || workspace_bytes < esmi_fuse_variance_adaptor_workspace_bytes(B, T, dim, depth)) return ESMI_ERR_WORKSPACE;
    const size_t fws = esmi_fuse_workspace_bytes(B, T, dim, depth);
    int rc = esmi_fuse_f32(fw, depth, dim, kernel, B, T, feats, n_i, mask, feat, 4 * dim, workspace, fws, stream);
    if (rc) return rc;
    rc = esmi_variance_adaptor_f32(pitch, energy, duration, dim, B, T, mask, pitch_target, energy_target,
                                   duration_target, feat, pitch_pred, energy_pred, duration_pred, pitch_idx, energy_idx,
                                   dur, static_cast<char*>(workspace) + fws, workspace_bytes - fws, stream);
    if (rc || !cum) return rc;
    ESMI_LAUNCH(length_regulate_kernel, dim3(B), dim3(64), 0, S(stream), dur, T, cum, mel_len, (int*)nullptr);
    return launch_status();
}

float esmi_split_weight_limit(void) {
#if ESMI_CHAIN_SPLIT || ESMI_DEC_SPLIT
    return 65504.0f / kF16WScale;   // 2^8 * W must stay a finite binary16 number
#else
    return __builtin_huge_valf();
#endif
}

int esmi_absmax_f32(const float* x, int64_t n, float* out, esmi_stream_t stream) {
    if (!x || !out || n <= 0) return ESMI_ERR_ARG;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float), S(stream));
    if (e != hipSuccess) return (int)e;
    const long blocks = (n + 256L * 8 - 1) / (256L * 8);
    ESMI_LAUNCH(absmax_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, S(stream), x, (long)n, reinterpret_cast<int*>(out));
    return launch_status();
}

// ------------------------------------------------------------------ module-level forwards (one kernel per reference op)
size_t esmi_self_attention_workspace_bytes(int B, int N, int C, int heads) {
    if (B <= 0 || N <= 0 || C <= 0 || heads <= 0) return 0;
    const size_t rows = (size_t)B * N;
    return align256(rows * 3 * heads * C * 4) + align256(rows * heads * C * 4);
}

int esmi_self_attention_f32(const float* qkv_w, const float* proj_w, const float* proj_b, int B, int N, int C, int heads,
                            const float* x, float* out, void* workspace, size_t workspace_bytes, esmi_stream_t stream) {
    if (!qkv_w || !proj_w || !proj_b || !x || !out || !workspace || B <= 0 || N <= 0 || C <= 0 || heads <= 0) return ESMI_ERR_ARG;
    if (workspace_bytes < esmi_self_attention_workspace_bytes(B, N, C, heads)) return ESMI_ERR_WORKSPACE;
    float* qkv = static_cast<float*>(workspace);
    float* ctx = reinterpret_cast<float*>(static_cast<char*>(workspace) + align256((size_t)B * N * 3 * heads * C * 4));
    hipStream_t st = S(stream);
    int rc;
    ConvGemmP p = conv_defaults();   // qkv Linear (bias-free), blocks.py:44
    p.B = B; p.n_in = N; p.c_in = C; p.n_out = N; p.c_out = 3 * heads * C;
    p.A = x; p.lda = C; p.W = qkv_w; p.out = qkv; p.ldo = 3 * heads * C;
    if ((rc = launch_convgemm(p, st))) return rc;
    AttnP a;                          // softmax(q k^T scale) v, blocks.py:49-64 (scores not masked)
    a.qkv = qkv; a.B = B; a.N = N; a.C = C; a.h = heads; a.ctx = ctx;
    a.scale = 1.0f / sqrtf((float)(C / heads));
    if ((rc = launch_attn(a, st))) return rc;
    p = conv_defaults();              // proj, blocks.py:65
    p.B = B; p.n_in = N; p.c_in = heads * C; p.n_out = N; p.c_out = C;
    p.A = ctx; p.lda = heads * C; p.W = proj_w; p.bias = proj_b; p.out = out; p.ldo = C;
    return launch_convgemm(p, st);
}

size_t esmi_mixffn_workspace_bytes(int B, int N, int C, int expansion) {
    if (B <= 0 || N <= 0 || C <= 0 || expansion <= 0) return 0;
    return 2 * align256((size_t)B * N * C * expansion * 4);
}

int esmi_mixffn_f32(const float* mlp1_w, const float* mlp1_b, const float* conv_w, const float* conv_b, const float* mlp2_w,
                    const float* mlp2_b, int B, int N, int C, int expansion, const float* x, float* out, void* workspace,
                    size_t workspace_bytes, esmi_stream_t stream) {
    if (!mlp1_w || !mlp1_b || !conv_w || !conv_b || !mlp2_w || !mlp2_b || !x || !out || !workspace) return ESMI_ERR_ARG;
    if (B <= 0 || N <= 0 || C <= 0 || expansion <= 0) return ESMI_ERR_ARG;
    if (workspace_bytes < esmi_mixffn_workspace_bytes(B, N, C, expansion)) return ESMI_ERR_WORKSPACE;
    const int E = C * expansion;
    float* m1 = static_cast<float*>(workspace);
    float* m2 = reinterpret_cast<float*>(static_cast<char*>(workspace) + align256((size_t)B * N * E * 4));
    hipStream_t st = S(stream);
    int rc;
    ConvGemmP p = conv_defaults();   // mlp1, blocks.py:23
    p.B = B; p.n_in = N; p.c_in = C; p.n_out = N; p.c_out = E;
    p.A = x; p.lda = C; p.W = mlp1_w; p.bias = mlp1_b; p.out = m1; p.ldo = E;
    if ((rc = launch_convgemm(p, st))) return rc;
    p = conv_defaults();              // dense k=3 conv + exact-erf GELU, blocks.py:24-27
    p.B = B; p.n_in = N; p.c_in = E; p.n_out = N; p.c_out = E; p.k = 3; p.pad = 1;
    p.A = m1; p.lda = E; p.W = conv_w; p.bias = conv_b; p.act = ACT_GELU; p.out = m2; p.ldo = E;
    if ((rc = launch_convgemm(p, st))) return rc;
    p = conv_defaults();              // mlp2, blocks.py:28
    p.B = B; p.n_in = N; p.c_in = E; p.n_out = N; p.c_out = C;
    p.A = m2; p.lda = E; p.W = mlp2_w; p.bias = mlp2_b; p.out = out; p.ldo = C;
    return launch_convgemm(p, st);
}

int esmi_acoustic_decoder_f32(const esmi_predictor_weights* w, int dim, int B, int T, int duration, const float* x, int ldx,
                              float* pred, float* features, void* workspace, size_t workspace_bytes, esmi_stream_t stream) {
    if (!w || !x || !pred || !workspace || dim <= 0 || B <= 0 || T <= 0 || ldx < dim) return ESMI_ERR_ARG;
    if (duration && !features) return ESMI_ERR_ARG;
    if (workspace_bytes < esmi_variance_adaptor_workspace_bytes(B, T, dim)) return ESMI_ERR_WORKSPACE;
    float* t1 = static_cast<float*>(workspace);
    hipStream_t st = S(stream);
    ConvGemmP p = conv_defaults();   // conv1 + ReLU -> LN1 -> ReLU, networks.py:152-155
    p.B = B; p.n_in = T; p.c_in = dim; p.n_out = T; p.c_out = dim; p.k = 3; p.pad = 1;
    p.A = x; p.lda = ldx; p.W = w->conv1_w; p.bias = w->conv1_b; p.act = ACT_RELU;
    p.ln_g = w->ln1_g; p.ln_b = w->ln1_b; p.post_relu = 1; p.out = t1; p.ldo = dim;
    int rc = launch_convgemm(p, st);
    if (rc) return rc;
    p = conv_defaults();              // conv2 + ReLU; y = Linear(dim,1) on the PRE-norm2 tensor (:157-160); duration: ReLU + features = LN2
    p.B = B; p.n_in = T; p.c_in = dim; p.n_out = T; p.c_out = dim; p.k = 3; p.pad = 1;
    p.A = t1; p.lda = dim; p.W = w->conv2_w; p.bias = w->conv2_b; p.act = ACT_RELU;
    p.dot_w = w->lin_w; p.dot_b = w->lin_b; p.dot_out = pred; p.dot_relu = duration != 0;
    if (duration) { p.ln_g = w->ln2_g; p.ln_b = w->ln2_b; p.out = features; p.ldo = dim; }
    return launch_convgemm(p, st);
}

int esmi_bucket_embedding_f32(const float* v, const float* bins, const float* emb, int64_t rows, int dim, float* out,
                              int32_t* idx, esmi_stream_t stream) {
    if (!v || !bins || !emb || !out || rows <= 0 || dim <= 1) return ESMI_ERR_ARG;
    const long n = (long)rows * dim;
    ESMI_LAUNCH(bucket_embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), v, bins, emb, (long)rows, dim, out, idx);
    return launch_status();
}

int esmi_max_i32(const int32_t* v, int n, int32_t* out, esmi_stream_t stream) {
    if (!v || !out || n <= 0) return ESMI_ERR_ARG;
    ESMI_LAUNCH(max_i32_kernel, dim3(1), dim3(64), 0, S(stream), v, n, out);
    return launch_status();
}

int esmi_length_regulate_i32(const int32_t* dur, int B, int T, int32_t* cum, int32_t* mel_len, int32_t* lmax,
                             esmi_stream_t stream) {
    if (!dur || !cum || !mel_len || !lmax || B <= 0 || T <= 0) return ESMI_ERR_ARG;
    hipError_t e = hipMemsetAsync(lmax, 0, sizeof(int32_t), S(stream));
    if (e != hipSuccess) return (int)e;
    ESMI_LAUNCH(length_regulate_kernel, dim3(B), dim3(64), 0, S(stream), dur, T, cum, mel_len, lmax);
    return launch_status();
}

int esmi_length_regulator_indices_i32(const int32_t* cum, int B, int T, int L, int32_t* idx, esmi_stream_t stream) {
    if (!cum || !idx || B <= 0 || T <= 0 || L <= 0) return ESMI_ERR_ARG;
    const long n = (long)B * L;
    ESMI_LAUNCH(lr_indices_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), cum, B, T, L, idx);
    return launch_status();
}

int esmi_upsample_f32(const float* feat, const uint8_t* fmask, const int32_t* cum, int B, int T, int C, int L,
                      float* features, uint8_t* masks, esmi_stream_t stream) {
    if (!feat || !cum || !features || B <= 0 || T <= 0 || L <= 0 || (C & 3)) return ESMI_ERR_ARG;
    const long n = (long)B * L * (C / 4);
    ESMI_LAUNCH(upsample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), feat, fmask, cum, B, T, C, L,
                features, masks);
    return launch_status();
}

int esmi_mask_rows_f32(float* x, const uint8_t* mask, int64_t rows, int C, esmi_stream_t stream) {
    if (!x || !mask || rows <= 0 || C <= 0) return ESMI_ERR_ARG;
    const long n = (long)rows * C;
    ESMI_LAUNCH(mask_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), x, mask, (long)rows, C);
    return launch_status();
}

// ------------------------------------------------------------------ mel decoder
static int dec_check(const esmi_decoder_shape* s) {
    if (!s) return ESMI_ERR_ARG;
    if (s->dx2 != 128 && s->dx2 != 256) return ESMI_ERR_UNSUPPORTED;
    if (s->d4 <= 0 || s->d4 % 128) return ESMI_ERR_UNSUPPORTED;
    if (s->kernel != 3 && s->kernel != 5) return ESMI_ERR_UNSUPPORTED;
    if (s->n_mel <= 0 || s->n_mel > kMelCols) return ESMI_ERR_UNSUPPORTED;
    if (s->n_blocks < 1 || s->block_depth < 1 || s->n_blocks * s->block_depth > ESMI_MAX_DEC_LAYERS) return ESMI_ERR_UNSUPPORTED;
    if (2 * (s->kernel / 2) * s->n_blocks * s->block_depth >= kDecRows - 32) return ESMI_ERR_UNSUPPORTED;
    return ESMI_OK;
}

size_t esmi_mel_decoder_blob_bytes(const esmi_decoder_shape* s) {
    if (dec_check(s)) return 0;
    return (size_t)dec_layout(s->d4, s->dx2, s->kernel, s->n_blocks, s->block_depth).total * sizeof(float);
}

int esmi_mel_decoder_pack_f32(const esmi_decoder_weights* w, const esmi_decoder_shape* s, float* blob,
                              esmi_stream_t stream) {
    int rc = dec_check(s);
    if (rc) return rc;
    if (!w || !blob) return ESMI_ERR_ARG;
    const DecLayout L = dec_layout(s->d4, s->dx2, s->kernel, s->n_blocks, s->block_depth);
    hipStream_t st = S(stream);
    const int dx2 = s->dx2, ntw = dx2 / 128;
    auto bslice = [&](const float* src, long off, int N, int K) {
#if ESMI_DEC_SPLIT == 2
        const long n = (long)(K / 128) * 4 * ntw * 8 * 2 * 256;
        ESMI_LAUNCH(pack_bslice2h_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src,
                    reinterpret_cast<unsigned*>(blob + off), N, K, ntw);
#else
        const long n = (long)(K / 128) * 4 * ntw * 16 * 256;
        ESMI_LAUNCH(pack_bslice_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, blob + off, N, K, ntw);
#endif
    };
    auto vec = [&](const float* src, long off, int n, int n_pad) {
        ESMI_LAUNCH(copy_pad_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, st, src, blob + off, n, n_pad);
    };
    bslice(w->proj_w, L.proj_w, dx2, s->d4);
    vec(w->proj_b, L.proj_b, dx2, dx2);
    vec(w->proj_ln_g, L.proj_g, dx2, dx2);
    vec(w->proj_ln_b, L.proj_beta, dx2, dx2);
    for (int l = 0; l < s->n_blocks * s->block_depth; ++l) {
        const long base = L.layer0 + (long)l * L.layer_stride;
        if (!w->dw_w[l] || !w->pw_w[l]) return ESMI_ERR_ARG;
        ESMI_LAUNCH(pack_dw_kernel, dim3((dx2 * s->kernel + 255) / 256), dim3(256), 0, st, w->dw_w[l], blob + base + L.l_dw,
                    dx2, s->kernel);
        vec(w->dw_b[l], base + L.l_dwb, dx2, dx2);
        bslice(w->pw_w[l], base + L.l_pw, dx2, dx2);
        vec(w->pw_b[l], base + L.l_pwb, dx2, dx2);
        vec(w->ln_g[l], base + L.l_g, dx2, dx2);
        vec(w->ln_b[l], base + L.l_b, dx2, dx2);
    }
    for (int b = 0; b < s->n_blocks; ++b) {
        vec(w->skip_g[b], L.skip0 + 2L * dx2 * b, dx2, dx2);
        vec(w->skip_b[b], L.skip0 + 2L * dx2 * b + dx2, dx2, dx2);
    }
    bslice(w->mel_w, L.mel_w, s->n_mel, dx2);
    vec(w->mel_b, L.mel_b, s->n_mel, dx2);
    if (L.rows0 >= 0) {   // the row-owner form's copy (mel_decoder_rows.h)
        auto afrag = [&](const float* src, long off, int N, int MT) {
            const long n = 8L * MT * 2 * 256;
            ESMI_LAUNCH(pack_rows_afrag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src,
                        reinterpret_cast<unsigned*>(blob + off), N, MT);
        };
        for (int l = 0; l < s->n_blocks * s->block_depth; ++l) {
            const long base = L.rows0 + (long)l * L.rows_layer_stride, pr = base + 16384;
            afrag(w->pw_w[l], base, dx2, 4);
            ESMI_LAUNCH(pack_dw_kernel, dim3((dx2 * s->kernel + 255) / 256), dim3(256), 0, st, w->dw_w[l], blob + pr, dx2, s->kernel);
            ESMI_LAUNCH(pack_rows_bias_kernel, dim3(2), dim3(64), 0, st, w->pw_w[l], w->dw_b[l], w->pw_b[l],
                        blob + pr + (long)s->kernel * dx2);
            vec(w->ln_g[l], pr + (long)(s->kernel + 1) * dx2, dx2, dx2);
            vec(w->ln_b[l], pr + (long)(s->kernel + 2) * dx2, dx2, dx2);
        }
        afrag(w->mel_w, L.rows_mel, s->n_mel, 3);
        ESMI_LAUNCH(pack_rows_padrow_kernel, dim3(1), dim3(64), 0, st, w->proj_b, w->proj_ln_g, w->proj_ln_b, blob + L.rows_pad);
    }
    return launch_status();
}

#ifndef ESMI_DEC_ROWS
#define ESMI_DEC_ROWS 0     // 1: dx2 = 128 decoders with the phoneme-rate head take the row-owner kernel (mel_decoder_rows.h) by default.
#endif                      // Measured on MI355X (tiny B=256 T=128): 0.316 ms vs 0.198 ms for the tile form, so 0
static int mel_decoder_launch(const float* blob, const esmi_decoder_shape* s, const float* x, const float* h0,
                              const int32_t* cum, const int32_t* mel_len, const int32_t* lmax_dev, int lmax_host, int apply_mask, int B, int T,
                              int L_out, float* mel, esmi_stream_t stream, bool rows) {
    int rc = dec_check(s);
    if (rc) return rc;
    if (!blob || (!x && !h0) || !mel || B <= 0 || L_out <= 0 || !aligned16(blob) || (x && !aligned16(x))) return ESMI_ERR_ARG;
    if (h0 && (!cum || !aligned16(h0))) return ESMI_ERR_ARG;   // the phoneme-rate head only exists in the fused-gather mode
    if (!cum && lmax_dev) return ESMI_ERR_ARG;  // direct mode: L is the tensor's own length, known to the host
    if (!lmax_dev && lmax_host == 0) return ESMI_ERR_ARG;
    if (!lmax_dev && lmax_host < 0 && (!mel_len || !cum)) return ESMI_ERR_ARG;   // L derived from mel_len
    MelDecP p;
    p.blob = blob;
    p.lay = dec_layout(s->d4, s->dx2, s->kernel, s->n_blocks, s->block_depth);
    p.d4 = s->d4; p.n_blocks = s->n_blocks; p.block_depth = s->block_depth; p.n_mel = s->n_mel;
    p.x = x; p.h0 = h0; p.cum = cum; p.mel_len = mel_len; p.lmax_dev = lmax_dev; p.lmax_host = lmax_host;
    p.apply_mask = apply_mask && mel_len; p.B = B; p.T = T; p.L_out = L_out; p.mel = mel;
    p.halo = (s->kernel / 2) * s->n_blocks * s->block_depth;
    p.TL = kDecRows - 2 * p.halo;
    p.trace = nullptr;
#ifdef ESMI_DEC_TRACE
    p.trace = g_esmi_trace;   // development only, see tools/trace_decoder.py
#endif
    hipStream_t st = S(stream);
    const bool rows_ok = h0 && p.lay.rows0 >= 0 && 4 * p.halo <= kRowsWin;
    if (rows && !rows_ok) return ESMI_ERR_UNSUPPORTED;
    if (rows) {
        p.TL = kRowsWin - 2 * p.halo;
        p.n_tiles = (L_out + p.TL - 1) / p.TL;
        dim3 rgrid((unsigned)(p.n_tiles * ((B + 7) / 8) * 8));
        const int lds = dec_rows_lds_floats(s->kernel) * (int)sizeof(float);
        if (s->kernel == 5) {
            static AttrOnce once;
            if (int rc2 = raise_lds_limit(reinterpret_cast<const void*>(mel_decoder_rows_kernel<5>), once)) return rc2;
            ESMI_LAUNCH((mel_decoder_rows_kernel<5>), rgrid, dim3(kRowsThreads), lds, st, p);
        } else {
            static AttrOnce once;
            if (int rc2 = raise_lds_limit(reinterpret_cast<const void*>(mel_decoder_rows_kernel<3>), once)) return rc2;
            ESMI_LAUNCH((mel_decoder_rows_kernel<3>), rgrid, dim3(kRowsThreads), lds, st, p);
        }
        return launch_status();
    }
    p.n_tiles = (L_out + p.TL - 1) / p.TL;
    dim3 grid((unsigned)(p.n_tiles * ((B + 7) / 8) * 8)), block(kDecThreads);
#ifndef ESMI_DEC_NW256
#define ESMI_DEC_NW256 8    // waves per window of the dx2 = 256 decoder (small / base ES): 8, or 16 (measured 30 % slower: 65 spilled
                            // VGPRs at the 128-register budget and twice the weight traffic; small ES decoder 2.43 vs 1.87 ms)
#endif
#define ESMI_DEC_CASE(DX2, KD, NW)                                                                                   \
    {                                                                                                              \
        const int lds = dec_lds_floats<DX2>(KD) * (int)sizeof(float);                                              \
        static AttrOnce once; /* per instantiation */                                                              \
        if (int rc2 = raise_lds_limit(reinterpret_cast<const void*>(mel_decoder_kernel<DX2, KD, NW>), once)) return rc2; \
        ESMI_LAUNCH((mel_decoder_kernel<DX2, KD, NW>), grid, dim3(64 * NW), lds, st, p);                           \
    }
    if (s->dx2 == 128 && s->kernel == 5) ESMI_DEC_CASE(128, 5, 8)
    else if (s->dx2 == 128 && s->kernel == 3) ESMI_DEC_CASE(128, 3, 8)
    else if (s->dx2 == 256 && s->kernel == 5) ESMI_DEC_CASE(256, 5, ESMI_DEC_NW256)
    else ESMI_DEC_CASE(256, 3, ESMI_DEC_NW256)
#undef ESMI_DEC_CASE
    return launch_status();
}

int esmi_mel_decoder_f32(const float* blob, const esmi_decoder_shape* s, const float* x, const float* h0,
                         const int32_t* cum, const int32_t* mel_len, const int32_t* lmax_dev, int lmax_host, int apply_mask, int B, int T,
                         int L_out, float* mel, esmi_stream_t stream) {
    const bool rows = ESMI_DEC_ROWS && h0 && s && s->dx2 == 128 && ESMI_DEC_SPLIT == 2 &&
                      4 * (s->kernel / 2) * s->n_blocks * s->block_depth <= kRowsWin;
    return mel_decoder_launch(blob, s, x, h0, cum, mel_len, lmax_dev, lmax_host, apply_mask, B, T, L_out, mel, stream, rows);
}
int esmi_mel_decoder_rows_f32(const float* blob, const esmi_decoder_shape* s, const float* x, const float* h0,
                              const int32_t* cum, const int32_t* mel_len, const int32_t* lmax_dev, int lmax_host, int apply_mask, int B,
                              int T, int L_out, float* mel, esmi_stream_t stream) {
    return mel_decoder_launch(blob, s, x, h0, cum, mel_len, lmax_dev, lmax_host, apply_mask, B, T, L_out, mel, stream, true);
}

// ------------------------------------------------------------------ HiFi-GAN generator
namespace {
struct HgPlan {
    size_t buf;      // bytes of one activation buffer (the largest B * N * C of the chain)
    long n_last;     // samples per utterance
};
int hg_plan(const esmi_hifigan_shape* s, int B, int L, HgPlan* o) {
    if (!s || B <= 0 || L <= 0 || s->n_up < 1 || s->n_up > ESMI_HIFIGAN_MAX_UP || s->n_kernels < 1 ||
        s->n_kernels > ESMI_HIFIGAN_MAX_KERNELS || (s->resblock != 1 && s->resblock != 2) || s->n_mel <= 0 || (s->n_mel & 7))
        return ESMI_ERR_ARG;
    if (s->n_up * s->n_kernels * 3 > ESMI_HIFIGAN_MAX_RBCONV) return ESMI_ERR_UNSUPPORTED;
    long n = L;
    int c = s->initial_channel;
    size_t mx = (size_t)B * n * c;
    for (int i = 0; i < s->n_up; ++i) {
        if (s->up_rates[i] < 1 || s->up_kernels[i] < s->up_rates[i] || ((s->up_kernels[i] - s->up_rates[i]) & 1) || (c & 15)) return ESMI_ERR_UNSUPPORTED;
        n *= s->up_rates[i];
        c /= 2;
        const size_t e = (size_t)B * n * c;
        mx = e > mx ? e : mx;
    }
    if (c & 7) return ESMI_ERR_UNSUPPORTED;   // implicit-GEMM k-steps are 8 channels
    o->buf = align256(mx * 4);
    o->n_last = n;
    return ESMI_OK;
}

bool resblock_fused_ok(const esmi_hifigan_weights* w, const esmi_hifigan_shape* s, int rb, int c, int k, int n, ResblockP* o) {
#if !ESMI_CHAIN_SPLIT
    return false;   // the exact-fp32 build keeps the per-conv fp32-MFMA launches
#endif
    if ((c != 8 && c != 16 && c != 32 && c != 64) || (k != 3 && k != 7 && k != 11)) return false;   // the instantiations
    const int nconv = s->resblock == 1 ? 3 : 2, j = rb % s->n_kernels;
    ResblockP p = {};
    int halo = 0, q = 0;
    for (int m = 0; m < nconv; ++m) {
        const int d = s->rb_dilations[j * 3 + m];
        if (d < 1 || !w->rb_wp1[rb * 3 + m] || !w->rb_b1[rb * 3 + m]) return false;
        p.conv[q++] = RbConv{static_cast<const unsigned*>(w->rb_wp1[rb * 3 + m]), w->rb_b1[rb * 3 + m], d, s->resblock == 1 ? 0 : 1};
        halo += (k - 1) / 2 * d;
        if (s->resblock == 1) {
            if (!w->rb_wp2[rb * 3 + m] || !w->rb_b2[rb * 3 + m]) return false;
            p.conv[q++] = RbConv{static_cast<const unsigned*>(w->rb_wp2[rb * 3 + m]), w->rb_b2[rb * 3 + m], 1, 1};
            halo += (k - 1) / 2;
        }
    }
    const int r_max = c == 64 ? 256 : 512;    // 8 waves = 8 (row pair, 32-channel tile) items; LDS <= 80 KB: two workgroups per CU
    int R = ((n + 2 * halo + 63) / 64) * 64;
    R = R < r_max ? R : r_max;
    if (R - 2 * halo < 32 && R - 2 * halo < n) return false;
    p.n_conv = q; p.k = k; p.halo = halo; p.R = R; p.TL = R - 2 * halo; p.n = n;
    p.tiles_per_b = (n + p.TL - 1) / p.TL;
    *o = p;
    return true;
}
}  // namespace

size_t esmi_hifigan_workspace_bytes(const esmi_hifigan_shape* s, int B, int L) {
    HgPlan o;
    return hg_plan(s, B, L, &o) == ESMI_OK ? 4 * o.buf : 0;
}

int esmi_hifigan_generator_f32(const esmi_hifigan_weights* w, const esmi_hifigan_shape* s, const float* mel, int B, int L,
                               float* wav, void* workspace, size_t workspace_bytes, esmi_stream_t stream) {
    HgPlan o;
    int rc = hg_plan(s, B, L, &o);
    if (rc) return rc;
    if (!w || !mel || !wav || !workspace) return ESMI_ERR_ARG;
    if (workspace_bytes < 4 * o.buf) return ESMI_ERR_WORKSPACE;
    hipStream_t st = S(stream);
    float* bufs[4];
    for (int q = 0; q < 4; ++q) bufs[q] = reinterpret_cast<float*>(static_cast<char*>(workspace) + q * o.buf);
    float *x = bufs[0], *y = bufs[1], *r = bufs[2], *t = bufs[3];   // x: stage input / sum over the ResBlocks; y: upsampled; r, t: ResBlock state
    const float slope = 0.1f;                                       // LRELU_SLOPE, hifigan/models.py:17
    long n = L;
    int c = s->initial_channel;
    ConvGemmP p = conv_defaults();   // conv_pre, models.py:112: Conv1d(n_mel, C0, 7, padding 3)
    p.B = B; p.n_in = L; p.c_in = s->n_mel; p.n_out = L; p.c_out = c; p.k = 7; p.pad = 3;
    p.A = mel; p.lda = s->n_mel; p.W = w->pre_w; p.bias = w->pre_b; p.out = x; p.ldo = c;
    if (!p.W || !p.bias) return ESMI_ERR_ARG;
    if ((rc = launch_convgemm(p, st))) return rc;
    float in_scale = 1.0f;           // the mean over the ResBlocks of the previous stage, folded into the next input activation
    for (int i = 0; i < s->n_up; ++i) {
        const int u = s->up_rates[i], k = s->up_kernels[i], co = c / 2;
        const long no = n * u;
        // x = ups[i](leaky_relu(x, 0.1)), models.py:114-115: ConvTranspose1d(c, c/2, k, u, padding (k-u)//2)
        p = conv_defaults();
        p.mode = MODE_CONVT; p.k = k; p.stride = u; p.pad = (k - u) / 2;
        p.B = B; p.n_in = (int)n; p.c_in = c; p.n_out = (int)no; p.c_out = co;
        p.A = x; p.lda = c; p.W = w->up_w[i]; p.bias = w->up_b[i]; p.out = y; p.ldo = co;
        p.act_in = 1; p.act_in_slope = slope; p.a_scale = in_scale;
        if (!p.W || !p.bias) return ESMI_ERR_ARG;
        if ((rc = launch_convgemm(p, st))) return rc;
        n = no; c = co;
        for (int j = 0; j < s->n_kernels; ++j) {   // xs += resblocks[i*num_kernels + j](x), models.py:116-121
            const int rb = i * s->n_kernels + j, kk = s->rb_kernels[j];
            const int nconv = s->resblock == 1 ? 3 : 2;
            ResblockP fp;
            if (resblock_fused_ok(w, s, rb, c, kk, (int)n, &fp)) {   // the whole block on an LDS-resident window: y -> x (+)=
                fp.x = y; fp.out = x; fp.B = B; fp.accum = j > 0; fp.slope = slope;
                if ((rc = launch_resblock(fp, c, st))) return rc;
                continue;
            }
            const float* cur = y;                   // the ResBlock's running x (first iteration: the stage input itself)
            for (int m = 0; m < nconv; ++m) {
                const int d = s->rb_dilations[j * 3 + m];
                const bool last = m + 1 == nconv;
                float* dst = last ? x : (cur == r ? t : r);   // last iteration: straight into the stage sum (accumulated for j > 0)
                if (s->resblock == 1) {
                    // xt = c1(leaky_relu(x)); xt = c2(leaky_relu(xt)); x = xt + x   (models.py:49-54)
                    // buffers: cur in {y, r}; c1 writes t; c2 reads t, adds cur, writes dst in {r (in place when cur == r), x}
                    p = conv_defaults();
                    p.B = B; p.n_in = (int)n; p.c_in = c; p.n_out = (int)n; p.c_out = c; p.k = kk; p.dil = d; p.pad = (kk * d - d) / 2;
                    p.A = cur; p.lda = c; p.W = w->rb_w1[rb * 3 + m]; p.bias = w->rb_b1[rb * 3 + m]; p.out = t; p.ldo = c;
                    p.act_in = 1; p.act_in_slope = slope;
                    if (!p.W || !p.bias) return ESMI_ERR_ARG;
                    if ((rc = launch_convgemm(p, st))) return rc;
                    p = conv_defaults();
                    p.B = B; p.n_in = (int)n; p.c_in = c; p.n_out = (int)n; p.c_out = c; p.k = kk; p.dil = 1; p.pad = (kk - 1) / 2;
                    p.A = t; p.lda = c; p.W = w->rb_w2[rb * 3 + m]; p.bias = w->rb_b2[rb * 3 + m];
                    p.res = cur; p.ldr = c;
                    p.out = last ? x : r; p.ldo = c; p.accum = last && j > 0;
                    p.act_in = 1; p.act_in_slope = slope;
                    if (!p.W || !p.bias) return ESMI_ERR_ARG;
                    if ((rc = launch_convgemm(p, st))) return rc;
                    cur = r;
                } else {
                    // xt = c(leaky_relu(x)); x = xt + x   (models.py:75-79): the conv reads neighbours of x, so not in place
                    p = conv_defaults();
                    p.B = B; p.n_in = (int)n; p.c_in = c; p.n_out = (int)n; p.c_out = c; p.k = kk; p.dil = d; p.pad = (kk * d - d) / 2;
                    p.A = cur; p.lda = c; p.W = w->rb_w1[rb * 3 + m]; p.bias = w->rb_b1[rb * 3 + m];
                    p.res = cur; p.ldr = c; p.out = dst; p.ldo = c; p.accum = last && j > 0;
                    p.act_in = 1; p.act_in_slope = slope;
                    if (!p.W || !p.bias) return ESMI_ERR_ARG;
                    if ((rc = launch_convgemm(p, st))) return rc;
                    cur = dst;
                }
            }
        }
        in_scale = 1.0f / (float)s->n_kernels;   // x = xs / num_kernels (models.py:122), applied where x is read next
    }
    // x = tanh(conv_post(leaky_relu(x))), models.py:123-125 (F.leaky_relu default slope 0.01)
    p = conv_defaults();
    p.B = B; p.n_in = (int)n; p.c_in = c; p.n_out = (int)n; p.c_out = 1; p.k = 7; p.pad = 3;
    p.A = x; p.lda = c; p.W = w->post_w; p.bias = w->post_b; p.out = wav; p.ldo = 1; p.act = ACT_TANH;
    p.act_in = 1; p.act_in_slope = 0.01f; p.a_scale = in_scale;
    if (!p.W || !p.bias) return ESMI_ERR_ARG;
    return launch_convgemm(p, st);
}

// ------------------------------------------------------------------ whole forward behind one call
namespace {
struct FwdArena {
    size_t feats[ESMI_MAX_DEPTH], ws, feat, preds[3], idx[2], dur, cum, h0, total;
    int n[ESMI_MAX_DEPTH];
};
int fwd_arena(const esmi_forward_args* a, FwdArena* o) {
    if (!a || a->depth < 1 || a->depth > ESMI_MAX_DEPTH || a->B <= 0 || a->T <= 0 || a->dim <= 0) return ESMI_ERR_ARG;
    size_t off = 0, ws = 0;
    int n_in = a->T;
    for (int i = 0; i < a->depth; ++i) {
        esmi_encoder_block_shape sh = a->shapes[i];
        sh.B = a->B; sh.n_in = n_in;
        const int n = conv_out_len(n_in, sh.kernel, sh.stride, sh.kernel / 2);
        if (n <= 0) return ESMI_ERR_ARG;
        o->n[i] = n;
        o->feats[i] = off; off += align256((size_t)a->B * n * sh.c_out * 4);
        const size_t w = esmi_encoder_block_workspace_bytes(&sh);
        ws = w > ws ? w : ws;
        n_in = n;
    }
    const size_t wv = esmi_fuse_variance_adaptor_workspace_bytes(a->B, a->T, a->dim, a->depth);
    ws = wv > ws ? wv : ws;
    const size_t rows = (size_t)a->B * a->T;
    o->ws = off; off += align256(ws);
    o->feat = off; off += align256(rows * 4 * a->dim * 4);
    for (int q = 0; q < 3; ++q) { o->preds[q] = off; off += align256(rows * 4); }
    for (int q = 0; q < 2; ++q) { o->idx[q] = off; off += align256(rows * 4); }
    o->dur = off; off += align256(rows * 4);
    o->cum = off; off += align256(rows * 4);
    o->h0 = off; off += a->head.proj_wp ? align256(rows * a->head.dx2 * 4) : 0;
    o->total = off;
    return ESMI_OK;
}
}  // namespace

size_t esmi_forward_arena_bytes(const esmi_forward_args* a) {
    FwdArena o;
    return fwd_arena(a, &o) == ESMI_OK ? o.total : 0;
}

int esmi_phoneme2mel_forward_f32(const esmi_forward_args* a, int stage, esmi_stream_t stream) {
    FwdArena o;
    int rc = fwd_arena(a, &o);
    if (rc) return rc;
    if (!a->ids || !a->mel_len || !a->duration_pred || !a->arena || stage < 0 || stage > 2) return ESMI_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(a->arena) & 15u) || a->arena_bytes < o.total) return ESMI_ERR_WORKSPACE;
    char* base = static_cast<char*>(a->arena);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    auto I = [&](size_t off) { return reinterpret_cast<int32_t*>(base + off); };
    const int B = a->B, T = a->T, plan = a->plan & ESMI_FUSE_ALL;
    float* feat = F(o.feat);
    int32_t* cum = a->cum ? a->cum : I(o.cum);
    float* h0 = a->head.proj_wp ? F(o.h0) : nullptr;
    const uint8_t* mask = a->mask;
    if (stage != 2) {
        const float* x_in = nullptr;
        int n_in = T;
        for (int i = 0; i < a->depth; ++i) {
            esmi_encoder_block_shape sh = a->shapes[i];
            sh.B = B; sh.n_in = n_in; sh.plan = plan;
            sh.mask_pool = 1; sh.mask_len = T;
            if (mask) {   // networks.py:69-70: pool = round(T / n), half to even
                const double r = (double)T / o.n[i];
                sh.mask_pool = (int)nearbyint(r);
                if ((T + sh.mask_pool - 1) / sh.mask_pool != o.n[i]) return ESMI_ERR_UNSUPPORTED;
            }
            rc = esmi_encoder_block_f32(&a->blocks[i], &sh, i == 0 ? a->ids : nullptr, i == 0 ? a->embed : nullptr, x_in, mask,
                                        F(o.feats[i]), base + o.ws, esmi_encoder_block_workspace_bytes(&sh), stream);
            if (rc) return rc;
            x_in = F(o.feats[i]);
            n_in = o.n[i];
        }
        const float* feats[ESMI_MAX_DEPTH];
        for (int i = 0; i < a->depth; ++i) feats[i] = F(o.feats[i]);
        float* pp = a->pitch_pred ? a->pitch_pred : F(o.preds[0]);
        float* ep = a->energy_pred ? a->energy_pred : F(o.preds[1]);
        int32_t* pi = a->pitch_idx ? a->pitch_idx : I(o.idx[0]);
        int32_t* ei = a->energy_idx ? a->energy_idx : I(o.idx[1]);
        int32_t* dur = a->dur ? a->dur : I(o.dur);
        const size_t wsb = esmi_fuse_variance_adaptor_workspace_bytes(B, T, a->dim, a->depth);
        const bool head_ok = fuse_va_head_ok(fuse_va_chain_ok(&a->fuse, a->depth, a->dim, a->fuse_kernel, o.n[0], T, &a->pitch, &a->energy,
                                                              &a->duration, plan), a->dim, &a->head);
        rc = esmi_fuse_variance_adaptor_f32(&a->fuse, a->depth, a->dim, a->fuse_kernel, B, T, feats, o.n, &a->pitch, &a->energy,
                                            &a->duration, mask, nullptr, nullptr, a->dur_forced, feat, pp, ep, a->duration_pred,
                                            pi, ei, dur, cum, a->mel_len, head_ok ? &a->head : nullptr, head_ok ? h0 : nullptr, plan,
                                            base + o.ws, wsb, stream);
        if (rc) return rc;
        if (a->lmax_dev && (rc = esmi_max_i32(a->mel_len, B, a->lmax_dev, stream))) return rc;
    }
    if (stage != 1) {
        if (a->L_out <= 0) return ESMI_OK;   // every duration zero: empty mel, nothing to launch
        if (!a->mel || !a->dec_blob) return ESMI_ERR_ARG;
        // (a stage-2 call re-derives whether stage 1 produced the phoneme-rate head: the same static test)
        const bool head_ok = fuse_va_head_ok(fuse_va_chain_ok(&a->fuse, a->depth, a->dim, a->fuse_kernel, o.n[0], T, &a->pitch, &a->energy,
                                                              &a->duration, plan), a->dim, &a->head);
        rc = esmi_mel_decoder_f32(a->dec_blob, &a->dec_shape, feat, head_ok ? h0 : nullptr, cum, a->mel_len,
                                  a->lmax_host < 0 ? a->lmax_dev : nullptr, a->lmax_host, mask != nullptr && B > 1, B, T, a->L_out,
                                  a->mel, stream);
        if (rc) return rc;
    }
    return ESMI_OK;
}

// ------------------------------------------------------------------ training step (csrc/train_ops.h)
namespace {
inline unsigned grid1d(long n, int block = 256) { return (unsigned)((n + block - 1) / block); }
int conv_desc_ok(const esmi_conv_desc* d, ConvDesc* o) {
    if (!d || d->B <= 0 || d->n_in <= 0 || d->n_out <= 0 || d->c_in <= 0 || d->c_out <= 0 || d->k <= 0 || d->stride <= 0 || d->pad < 0 ||
        d->groups <= 0)
        return ESMI_ERR_ARG;
    if (d->groups != 1 && (d->transposed || d->c_in % d->groups || d->c_out % d->groups)) return ESMI_ERR_UNSUPPORTED;
    *o = ConvDesc{d->B, d->n_in, d->c_in, d->n_out, d->c_out, d->k, d->stride, d->pad, d->groups, d->transposed ? 1 : 0};
    return ESMI_OK;
}
}  // namespace

namespace {
inline bool wgrad_depthwise(const ConvDesc& c) { return !c.transposed && c.groups == c.c_in && c.c_in == c.c_out && c.k <= 8; }
inline bool wgrad_on_mfma(const ConvDesc& c);
inline int wgrad_chunk(const ConvDesc& c) {   // rows per partial sum: fewer for small weights, whose parallelism must come from the chunks
    if (wgrad_on_mfma(c)) return kTrainChunkMfma;
    if (wgrad_depthwise(c)) return kTrainChunkDw;
    const long nw = (long)(c.transposed ? c.c_in * c.c_out : c.c_out * (c.c_in / c.groups)) * c.k;
    return nw < 1024 ? 32 : kTrainChunk;
}
inline bool wgrad_on_mfma(const ConvDesc& c) { return c.groups == 1 && c.c_in >= 8 && c.c_out >= 8 && (c.c_out & 3) == 0; }
}
// Dense convolutions (groups == 1) of the training step run on the matrix pipe through the inference path's implicit GEMM
// (convgemm.h) when the caller gives scratch for the tap-major copy of the weight: the forward as it is, the data gradient as
// the transposed problem -- d(Conv1d) is a ConvTranspose1d of dy with the same (Cout, Cin, k) tensor read as (Cin', Cout', k),
// d(ConvTranspose1d) is a Conv1d of dy with (Cin, Cout, k) read as (Cout', Cin', k).  Everything else (depthwise, channel
// counts the GEMM does not take, no scratch) runs the one-thread-per-element kernels of train_ops.h.
size_t esmi_train_conv_workspace_bytes(const esmi_conv_desc* d) {
    ConvDesc c;
    if (conv_desc_ok(d, &c) || c.groups != 1) return 0;
    return align256((size_t)c.k * c.c_out * c.c_in * sizeof(float)) + 256;   // + the data gradient's absmax / scale slots
}
namespace {
// one of the two implicit-GEMM problems of a dense conv: returns ESMI_ERR_UNSUPPORTED when the GEMM does not take the shape
int train_conv_gemm(const ConvDesc& c, bool grad, const float* in, const float* w, const float* bias, float* out, float* wt,
                    hipStream_t st) {
    const int cin = grad ? c.c_out : c.c_in, cout = grad ? c.c_in : c.c_out;      // of the GEMM problem
    if (c.groups != 1 || (cin & 7) || !wt) return ESMI_ERR_UNSUPPORTED;
    if (cout == 1 && (grad != (c.transposed != 0) || c.stride != 1)) return ESMI_ERR_UNSUPPORTED;   // the one-channel kernel is a plain conv
    const long n = (long)c.k * c.c_out * c.c_in;
    // tap-major (k, cout, cin) of the problem: forward conv / grad of convT read the tensor as Conv1d, the other two as ConvTranspose1d
    const int as_convT = (grad != (c.transposed != 0)) ? 1 : 0;
    int* amax = reinterpret_cast<int*>(reinterpret_cast<char*>(wt) + align256((size_t)n * sizeof(float)));
    const float* wuse = wt;
    if (c.k == 1 && !as_convT && !grad) {
        wuse = w;                                   // a Linear's (Cout, Cin) IS its tap-major form: no copy
    } else {
        ESMI_LAUNCH(pack_conv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, wt, cout, cin, c.k, as_convT, grad ? amax : nullptr);
        if (int rc = launch_status()) return rc;
    }
    ConvGemmP p = conv_defaults();
    if (grad) {   // max|dy| on the device; the GEMM kernels derive the power-of-two scales from it (convgemm.h conv_pow2_scales)
        const long len = (long)c.B * c.n_out * c.c_out, blocks = (len + 256L * 8 - 1) / (256L * 8);
        ESMI_LAUNCH(absmax_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, st, in, len, amax);
        if (int rc = launch_status()) return rc;
        p.io_scale = reinterpret_cast<const float*>(amax);
    }
    p.mode = as_convT ? MODE_CONVT : MODE_CONV;
    p.k = c.k; p.stride = c.stride; p.pad = c.pad;
    p.B = c.B; p.n_in = grad ? c.n_out : c.n_in; p.n_out = grad ? c.n_in : c.n_out; p.c_in = cin; p.c_out = cout;
    p.A = in; p.lda = cin; p.W = wuse; p.bias = bias; p.out = out; p.ldo = cout;
    return launch_convgemm(p, st);
}
}  // namespace

int esmi_train_conv_fwd_f32(const esmi_conv_desc* d, const float* x, const float* w, const float* bias, float* y, void* workspace,
                            size_t workspace_bytes, esmi_stream_t stream) {
    ConvDesc c;
    if (int rc = conv_desc_ok(d, &c)) return rc;
    if (!x || !w || !y) return ESMI_ERR_ARG;
    if (workspace && workspace_bytes >= esmi_train_conv_workspace_bytes(d)) {
        const int rc = train_conv_gemm(c, false, x, w, bias, y, static_cast<float*>(workspace), S(stream));
        if (rc != ESMI_ERR_UNSUPPORTED) return rc;
    }
    const long n = (long)c.B * c.n_out * c.c_out;
    if (wgrad_depthwise(c) && c.stride == 1 && (c.c_out & 3) == 0) {
        ESMI_LAUNCH(train_conv_dw_kernel, grid1d(n / 4), dim3(256), 0, S(stream), c, x, w, bias, y, 0);
        return launch_status();
    }
    ESMI_LAUNCH(train_conv_fwd_kernel, grid1d(n), dim3(256), 0, S(stream), c, x, w, bias, y);
    return launch_status();
}
int esmi_train_conv_dgrad_f32(const esmi_conv_desc* d, const float* dy, const float* w, float* dx, void* workspace,
                              size_t workspace_bytes, esmi_stream_t stream) {
    ConvDesc c;
    if (int rc = conv_desc_ok(d, &c)) return rc;
    if (!dy || !w || !dx) return ESMI_ERR_ARG;
    if (workspace && workspace_bytes >= esmi_train_conv_workspace_bytes(d)) {
        const int rc = train_conv_gemm(c, true, dy, w, nullptr, dx, static_cast<float*>(workspace), S(stream));
        if (rc != ESMI_ERR_UNSUPPORTED) return rc;
    }
    const long n = (long)c.B * c.n_in * c.c_in;
    if (wgrad_depthwise(c) && c.stride == 1 && (c.c_out & 3) == 0) {
        ESMI_LAUNCH(train_conv_dw_kernel, grid1d(n / 4), dim3(256), 0, S(stream), c, dy, w, nullptr, dx, 1);
        return launch_status();
    }
    ESMI_LAUNCH(train_conv_dgrad_kernel, grid1d(n), dim3(256), 0, S(stream), c, dy, w, dx);
    return launch_status();
}
size_t esmi_train_conv_wgrad_workspace_bytes(const esmi_conv_desc* d) {
    ConvDesc c;
    if (conv_desc_ok(d, &c)) return 0;
    const long nw = (long)(c.transposed ? c.c_in * c.c_out : c.c_out * (c.c_in / c.groups)) * c.k;
    const long chunks = train_chunks((long)c.B * c.n_out, wgrad_chunk(c));
    return (size_t)chunks * (size_t)(nw + c.c_out) * sizeof(float);
}
int esmi_train_conv_wgrad_f32(const esmi_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias, void* workspace,
                              size_t workspace_bytes, esmi_stream_t stream) {
    ConvDesc c;
    if (int rc = conv_desc_ok(d, &c)) return rc;
    if (!x || !dy || !dw || !workspace) return ESMI_ERR_ARG;
    if (workspace_bytes < esmi_train_conv_wgrad_workspace_bytes(d)) return ESMI_ERR_WORKSPACE;
    const long nw = (long)(c.transposed ? c.c_in * c.c_out : c.c_out * (c.c_in / c.groups)) * c.k;
    const bool mfma = wgrad_on_mfma(c), depthwise = wgrad_depthwise(c);
    const long rows = (long)c.B * c.n_out, chunks = train_chunks(rows, wgrad_chunk(c));
    float* part = static_cast<float*>(workspace);   // [chunk][weight partials (nw) | bias partials (c_out)]
    float* pb = part + nw;
    const long ps = nw + c.c_out;
    if (mfma) {   // dense: one wave per (128 output channels x 32 input channels, tap, chunk) on the fp32 MFMA; bias partials from the tap 0, ci0 = 0 waves
        const unsigned tiles = (unsigned)(((c.c_out + 127) / 128) * ((c.c_in + 31) / 32) * c.k);
        ESMI_LAUNCH(train_conv_wgrad_mfma_kernel, dim3(tiles, (unsigned)((chunks + 3) / 4)), dim3(256), 0, S(stream), c, x, dy, part,
                    dbias ? pb : nullptr, chunks, ps);
    } else if (depthwise) {
        ESMI_LAUNCH(train_conv_wgrad_dw_kernel, dim3(grid1d(c.c_out, 64), (unsigned)chunks), dim3(64), 0, S(stream), c, x, dy, part,
                    dbias ? pb : nullptr, ps);
    } else {
        ESMI_LAUNCH(train_conv_wgrad_kernel, dim3(grid1d(nw, 64), (unsigned)chunks), dim3(64), 0, S(stream), c, x, dy, part, ps, wgrad_chunk(c));
        if (int rc = launch_status()) return rc;
        if (dbias) ESMI_LAUNCH(train_colsum_kernel, dim3(grid1d(c.c_out, 64), (unsigned)chunks), dim3(64), 0, S(stream), dy, rows, c.c_out, pb, ps, wgrad_chunk(c));
    }
    if (int rc = launch_status()) return rc;
    // weight and bias partials in ONE reduction launch: elements >= nw of a partial row are the bias sums
    ESMI_LAUNCH(train_reduce_chunks_kernel, grid1d(dbias ? ps : nw, 64), dim3(64 * kReduceGroups), 64 * kReduceGroups * sizeof(float), S(stream),
                part, dbias ? ps : nw, ps, chunks, dw, nw, dbias);
    return launch_status();
}
int esmi_train_layernorm_fwd_f32(const float* x, const float* g, const float* b, int64_t rows, int C, float* y, float* mean,
                                 float* rstd, esmi_stream_t stream) {
    if (!x || !g || !b || !y || !mean || !rstd || rows <= 0 || C <= 0) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_ln_fwd_kernel, grid1d(rows, 4), dim3(256), 0, S(stream), x, g, b, (long)rows, C, 1e-5f, y, mean, rstd);
    return launch_status();
}
size_t esmi_train_layernorm_bwd_workspace_bytes(int64_t rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    return (size_t)train_chunks(rows, C <= 256 ? kLnRows : kTrainChunk) * 2 * C * sizeof(float);
}
int esmi_train_layernorm_bwd_f32(const float* x, const float* g, const float* mean, const float* rstd, const float* dy,
                                 int64_t rows, int C, float* dx, float* dg, float* db, void* workspace, size_t workspace_bytes,
                                 esmi_stream_t stream) {
    if (!x || !g || !mean || !rstd || !dy || !dx || !dg || !db || !workspace || rows <= 0 || C <= 0) return ESMI_ERR_ARG;
    if (workspace_bytes < esmi_train_layernorm_bwd_workspace_bytes(rows, C)) return ESMI_ERR_WORKSPACE;
    float* part = static_cast<float*>(workspace);
    long chunks;
    if (C <= 256) {   // dx and the parameter partials in one pass
        chunks = train_chunks(rows, kLnRows);
        ESMI_LAUNCH(train_ln_bwd_fused_kernel, dim3((unsigned)chunks), dim3(256), 4 * 2 * 256 * sizeof(float), S(stream), x, g, mean, rstd, dy, (long)rows, C, dx, part);
        if (int rc = launch_status()) return rc;
    } else {
        chunks = train_chunks(rows);
        ESMI_LAUNCH(train_ln_bwd_dx_kernel, grid1d(rows, 4), dim3(256), 0, S(stream), x, g, mean, rstd, dy, (long)rows, C, dx);
        if (int rc = launch_status()) return rc;
        ESMI_LAUNCH(train_ln_bwd_params_kernel, dim3(grid1d(C, 64), (unsigned)chunks), dim3(64), 0, S(stream), x, mean, rstd, dy, (long)rows, C, part);
        if (int rc = launch_status()) return rc;
    }
    ESMI_LAUNCH(train_reduce_chunks_kernel, grid1d(2L * C, 64), dim3(64 * kReduceGroups), 64 * kReduceGroups * sizeof(float), S(stream), part,
                2L * C, 2L * C, chunks, dg, (long)C, db);   // partial rows are [dg (C) | db (C)]: one launch, two outputs
    return launch_status();
}
int esmi_train_act_fwd_f32(const float* x, int64_t n, int kind, float* y, esmi_stream_t stream) {
    if (!x || !y || n <= 0 || kind < ACT_RELU || kind > ACT_TANH) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_act_fwd_kernel, grid1d(n), dim3(256), 0, S(stream), x, (long)n, kind, y);
    return launch_status();
}
int esmi_train_act_bwd_f32(const float* saved, const float* dy, int64_t n, int kind, float* dx, esmi_stream_t stream) {
    if (!saved || !dy || !dx || n <= 0 || kind < ACT_RELU || kind > ACT_TANH) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_act_bwd_kernel, grid1d(n), dim3(256), 0, S(stream), saved, dy, (long)n, kind, dx);
    return launch_status();
}
int esmi_train_attention_fwd_f32(const float* qkv, int B, int N, int C, int h, float* P, float* ctx, esmi_stream_t stream) {
    if (!qkv || !P || !ctx || B <= 0 || N <= 0 || C <= 0 || h <= 0 || C % h) return ESMI_ERR_ARG;
    const size_t lds = train_attn_lds_bytes(N, C);
    if (lds <= 150 * 1024) {   // K and V of a head staged in LDS
        static AttrOnce once;
        if (lds > 48 * 1024)
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(train_attn_fwd_lds_kernel), once)) return rc;
        ESMI_LAUNCH(train_attn_fwd_lds_kernel, dim3((unsigned)(B * h)), dim3(256), lds, S(stream), qkv, B, N, C, h, 1.0f / sqrtf((float)(C / h)), P, ctx);
        return launch_status();
    }
    ESMI_LAUNCH(train_attn_fwd_kernel, dim3((unsigned)((long)B * h * N)), dim3(64), 0, S(stream), qkv, B, N, C, h, 1.0f / sqrtf((float)(C / h)), P, ctx);
    return launch_status();
}
int esmi_train_attention_bwd_f32(const float* qkv, const float* P, const float* dctx, int B, int N, int C, int h, float* dS,
                                 float* dqkv, esmi_stream_t stream) {
    if (!qkv || !P || !dctx || !dS || !dqkv || B <= 0 || N <= 0 || C <= 0 || h <= 0 || C % h) return ESMI_ERR_ARG;
    const float scale = 1.0f / sqrtf((float)(C / h));
    const size_t lds = train_attn_lds_bytes(N, C);
    if (lds <= 150 * 1024) {
        static AttrOnce once;
        if (lds > 48 * 1024)
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(train_attn_bwd_rows_lds_kernel), once)) return rc;
        ESMI_LAUNCH(train_attn_bwd_rows_lds_kernel, dim3((unsigned)(B * h)), dim3(256), lds, S(stream), qkv, P, dctx, B, N, C, h, scale, dS, dqkv);
    } else {
        ESMI_LAUNCH(train_attn_bwd_rows_kernel, dim3((unsigned)((long)B * h * N)), dim3(64), 0, S(stream), qkv, P, dctx, B, N, C, h, scale, dS, dqkv);
    }
    if (int rc = launch_status()) return rc;
    ESMI_LAUNCH(train_attn_bwd_cols_kernel, grid1d((long)B * h * N * C), dim3(256), 0, S(stream), qkv, P, dS, dctx, B, N, C, h, scale, dqkv);
    return launch_status();
}
int esmi_train_embedding_fwd_f32(const int32_t* ids, const float* table, int64_t rows, int V, int C, float* out, esmi_stream_t stream) {
    if (!ids || !table || !out || rows <= 0 || V <= 0 || C <= 0) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_embed_fwd_kernel, grid1d(rows * C), dim3(256), 0, S(stream), ids, table, (long)rows, V, C, out);
    return launch_status();
}
size_t esmi_train_embedding_bwd_workspace_bytes(int64_t rows, int V, int C) {
    return rows > 0 && V > 0 && C > 0 ? (size_t)train_chunks(rows) * V * C * sizeof(float) : 0;
}
int esmi_train_embedding_bwd_f32(const int32_t* ids, const float* dy, int64_t rows, int V, int C, int padding_idx, float* dtable,
                                 void* workspace, size_t workspace_bytes, esmi_stream_t stream) {
    if (!ids || !dy || !dtable || !workspace || rows <= 0 || V <= 0 || C <= 0) return ESMI_ERR_ARG;
    if (workspace_bytes < esmi_train_embedding_bwd_workspace_bytes(rows, V, C)) return ESMI_ERR_WORKSPACE;
    const long chunks = train_chunks(rows), n = (long)V * C;
    float* part = static_cast<float*>(workspace);
    ESMI_LAUNCH(train_embed_bwd_kernel, dim3(grid1d(n, 64), (unsigned)chunks), dim3(64), 0, S(stream), ids, dy, (long)rows, V, C, padding_idx, part);
    if (int rc = launch_status()) return rc;
    ESMI_LAUNCH(train_reduce_chunks_kernel, grid1d(n, 64), dim3(64 * kReduceGroups), 64 * kReduceGroups * sizeof(float), S(stream), part, n, n, chunks, dtable);
    return launch_status();
}
int esmi_train_mask_rows_f32(const float* x, const uint8_t* mask, int64_t rows, int C, float* y, esmi_stream_t stream) {
    if (!x || !mask || !y || rows <= 0 || C <= 0) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_mask_rows_kernel, grid1d(rows * C), dim3(256), 0, S(stream), x, mask, (long)rows, C, y);
    return launch_status();
}
int esmi_train_add_f32(const float* a, const float* b, int64_t n, float* y, esmi_stream_t stream) {
    if (!a || !b || !y || n <= 0) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_add_kernel, grid1d(n), dim3(256), 0, S(stream), a, b, (long)n, y);
    return launch_status();
}
int esmi_train_copy_cols_f32(const float* src, int ld_src, int col_src, float* dst, int ld_dst, int col_dst, int64_t rows, int C,
                             esmi_stream_t stream) {
    if (!src || !dst || rows <= 0 || C <= 0 || col_src < 0 || col_dst < 0 || col_src + C > ld_src || col_dst + C > ld_dst) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_copy_cols_kernel, grid1d(rows * C), dim3(256), 0, S(stream), src, ld_src, col_src, dst, ld_dst, col_dst, (long)rows, C);
    return launch_status();
}
int esmi_train_repeat_fwd_f32(const float* feat, const int32_t* cum, int B, int T, int C, int L, float* out, esmi_stream_t stream) {
    if (!feat || !cum || !out || B <= 0 || T <= 0 || C <= 0 || L <= 0) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_repeat_fwd_kernel, grid1d((long)B * L * C), dim3(256), 0, S(stream), feat, cum, B, T, C, L, out);
    return launch_status();
}
int esmi_train_repeat_bwd_f32(const float* dout, const int32_t* cum, int B, int T, int C, int L, float* dfeat, esmi_stream_t stream) {
    if (!dout || !cum || !dfeat || B <= 0 || T <= 0 || C <= 0 || L <= 0) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_repeat_bwd_kernel, grid1d((long)B * T * C), dim3(256), 0, S(stream), dout, cum, B, T, C, L, dfeat);
    return launch_status();
}
int esmi_train_loss_f32(const esmi_train_loss_args* a, esmi_stream_t stream) {
    if (!a || !a->mel_pred || !a->mel || !a->pitch_pred || !a->pitch || !a->energy_pred || !a->energy || !a->dur_pred || !a->dur ||
        !a->out || !a->d_mel || !a->d_pitch || !a->d_energy || !a->d_dur || !a->scratch || a->B <= 0 || a->T <= 0 || a->L <= 0 || a->n_mel <= 0)
        return ESMI_ERR_ARG;
    static_assert(ESMI_TRAIN_LOSS_SCRATCH_FLOATS >= kLossBlocks * 6, "scratch size in the header");
    LossP p = {a->mel_pred, a->mel, a->pitch_pred, a->pitch, a->energy_pred, a->energy, a->dur_pred, a->dur, a->mel_mask, a->ph_mask,
               a->B, a->T, a->L, a->n_mel, a->out, a->d_mel, a->d_pitch, a->d_energy, a->d_dur, a->scratch};
    ESMI_LAUNCH(train_loss_partial_kernel, dim3(kLossBlocks), dim3(256), 256 * sizeof(float), S(stream), p);
    if (int rc = launch_status()) return rc;
    ESMI_LAUNCH(train_loss_final_kernel, dim3(1), dim3(256), 256 * sizeof(float), S(stream), p);
    if (int rc = launch_status()) return rc;
    const long nm = (long)a->B * a->L * a->n_mel, np_ = (long)a->B * a->T;
    ESMI_LAUNCH(train_loss_grad_kernel, grid1d(nm > np_ ? nm : np_), dim3(256), 0, S(stream), p);
    return launch_status();
}
int esmi_train_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                         double weight_decay, int step, esmi_stream_t stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return ESMI_ERR_ARG;
    // 1 - beta^step in double (-expm1(step * log beta)): in fp32, 1 - 0.999f^t carries ~6e-5 relative error at small t
    const double bc1 = -expm1((double)step * log(beta1)), bc2 = -expm1((double)step * log(beta2));
    AdamWScalars h = {(float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)(1.0 - lr * weight_decay),
                      (float)(lr / bc1), (float)sqrt(bc2)};
    ESMI_LAUNCH(train_adamw_kernel, grid1d(n), dim3(256), 0, S(stream), p, g, m, v, (long)n, h);
    return launch_status();
}

int esmi_train_adamw_graph_f32(float* p, const float* g, float* m, float* v, int64_t n, float* hyper_dev, double beta1, double beta2,
                               double eps, double weight_decay, int32_t* step_dev, esmi_stream_t stream) {
    if (!p || !g || !m || !v || !hyper_dev || !step_dev || n <= 0) return ESMI_ERR_ARG;
    ESMI_LAUNCH(train_bump_step_kernel, dim3(1), dim3(64), 0, S(stream), step_dev, hyper_dev, beta1, beta2, weight_decay);
    if (int rc = launch_status()) return rc;
    AdamWScalars h = {(float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, 0.0f, 0.0f, 0.0f};
    ESMI_LAUNCH(train_adamw_dev_kernel, grid1d(n), dim3(256), 0, S(stream), p, g, m, v, (long)n, h, (const float*)hyper_dev);
    return launch_status();
}

}  // extern "C"
