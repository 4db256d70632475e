// esmi C-ABI: the host-side launch sequences behind include/esmi.h.
// Built by `hipcc --offload-arch=gfx950` into libesmi.so (product) and, unchanged, by the host
// clang++ with -DESMI_WAVESIM into libesmi_sim.so (CPU wave simulator used only by tests).
#include "launch.h"
#include "mel_decoder.h"   // esmi_decoder_shape helpers used by the one-call forward
#include "enc_ffn64.h"

using namespace esmi;
ESMI_TU_RANGE_SETTER(abi)

namespace {

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
// ... or can the stage run as its own phoneme-rate GEMM launch (esmi_decoder_head_f32) behind the variance adaptor?
bool head_gemm_ok(const esmi_decoder_head* head) {
    return head && head->proj_w && head->proj_b && head->ln_g && head->ln_b && head->d4 > 0 && (head->d4 & 7) == 0 &&
           (head->dx2 == 32 || head->dx2 == 64 || head->dx2 == 128 || head->dx2 == 256);
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

#ifdef ESMI_CHAIN_TRACE   // development: tools/trace_chain.py
extern "C" {
void esmi_dev_set_chain_trace_enc_attn_ffn(long long*);
void esmi_dev_set_chain_trace_enc_block(long long*);
void esmi_dev_set_chain_trace_enc_fuse_va(long long*);
void esmi_dev_set_chain_trace_enc_merge(long long*);
void esmi_dev_set_chain_trace(long long* ptr) {
    esmi_dev_set_chain_trace_enc_attn_ffn(ptr); esmi_dev_set_chain_trace_enc_block(ptr);
    esmi_dev_set_chain_trace_enc_fuse_va(ptr); esmi_dev_set_chain_trace_enc_merge(ptr);
}
}
#endif

extern "C" {

int esmi_version(void) { return ESMI_VERSION; }
const char* esmi_backend(void) { return kBackendName; }

const char* esmi_build_config(void) {
#if ESMI_CHAIN_SPLIT
#define ESMI_CFG_ENC_ ",enc_gemm=split-f16x2"
#else
#define ESMI_CFG_ENC_ ",enc_gemm=fp32-mfma"
#endif
#if ESMI_RANGE_CHECK
#define ESMI_CFG_RC_ ",range_check=1"
#else
#define ESMI_CFG_RC_ ""
#endif
    // ... and the compiler the library was built with: two of the chain kernels' forms are work-arounds validated on ROCm 7.2.0's hipcc
    // (chain16.h swap16_f: the select form of v_permlane16_swap; the DPP / ds_swizzle scheduling fence of the fallback LayerNorm,
    // profiles/r05_probes/fuse_va_wrong_rows.md) -- _lib.load() warns when this differs from the validated version
#define ESMI_STR2_(x) #x
#define ESMI_STR_(x) ESMI_STR2_(x)
#if defined(__clang_major__)
#define ESMI_CFG_CC_ ",clang=" ESMI_STR_(__clang_major__) "." ESMI_STR_(__clang_minor__) "." ESMI_STR_(__clang_patchlevel__)
#else
#define ESMI_CFG_CC_ ""
#endif
#if defined(HIP_VERSION_MAJOR)
#define ESMI_CFG_HIP_ ",hip=" ESMI_STR_(HIP_VERSION_MAJOR) "." ESMI_STR_(HIP_VERSION_MINOR) "." ESMI_STR_(HIP_VERSION_PATCH)
#else
#define ESMI_CFG_HIP_ ""
#endif
#if ESMI_DEC_SPLIT == 2
    return "dec_gemm=split-f16x2" ESMI_CFG_ENC_ ESMI_CFG_RC_ ESMI_CFG_HIP_ ESMI_CFG_CC_;
#else
    return "dec_gemm=fp32-mfma" ESMI_CFG_ENC_ ESMI_CFG_RC_ ESMI_CFG_HIP_ ESMI_CFG_CC_;
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

// The parameters of a whole encoder block as ONE folded chain-kernel launch (what esmi_encoder_block_f32 below hands
// launch_enc_block16 / launch_enc_block), for the one-launch encoder side of the one-call forward; false when the block's packed / folded
// weights are not all there.
static bool block_chain_params(const esmi_encoder_block_weights* w, const esmi_encoder_block_shape* s, const int32_t* ids, const float* embed,
                               const float* x_in, const uint8_t* mask, float* x_out, EncAttnFfnP* out) {
    const bool ffn_folded = w->ffn_cw && w->ffn_cwp && w->ffn_cb && w->ffn_cb_first && w->ffn_cb_last;
    if (!(w->merge_cwp && w->qkv_wp && w->proj_wp && ffn_folded && w->mlp2_wp && w->qk_wp && w->vo_wp)) return false;
    const int n = conv_out_len(s->n_in, s->kernel, s->stride, s->kernel / 2);
    EncAttnFfnP f;
    memset(&f, 0, sizeof f);
    EncMergeP& m = f.m;
    m.ids = ids; m.table = embed; m.vocab = s->vocab; m.x_in = ids ? nullptr : x_in;
    m.B = s->B; m.n_in = s->n_in; m.n_out = n; m.k = s->kernel; m.stride = s->stride; m.pad = s->kernel / 2; m.h = s->heads;
    m.merge_w = w->merge_cwp; m.qkv_w = w->qk_wp; m.emb_conv = ids ? w->emb_conv : nullptr; m.tiles_per_b = (n + 31) / 32;
    f.B = s->B; f.N = n; f.C = s->c_out; f.h = s->heads; f.scale = 1.0f / sqrtf((float)(s->c_out / s->heads));
    f.proj_w = w->vo_wp; f.proj_b = w->proj_b; f.ln1_g = w->ln1_g; f.ln1_b = w->ln1_b;
    f.ffn_w = w->ffn_cwp; f.ffn_b = w->ffn_cb; f.ffn_b0 = w->ffn_cb_first; f.ffn_b2 = w->ffn_cb_last;
    f.mlp2_w = w->mlp2_wp; f.mlp2_b = w->mlp2_b; f.ln2_g = w->ln2_g; f.ln2_b = w->ln2_b;
    f.mask = mask; f.out = x_out;
    f.mask_pool = s->mask_pool > 0 ? s->mask_pool : 1; f.mask_len = s->mask_pool > 0 ? s->mask_len : n;
    f.fold = 1; f.wgs_per_b = 1; f.halo = 0;
    *out = f;
    return true;
}

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
    const bool ffn_folded = w->ffn_cw && w->ffn_cwp && w->ffn_cb && w->ffn_cb_first && w->ffn_cb_last;
    const bool packed = w->merge_cwp && w->qkv_wp && w->proj_wp && ffn_folded && w->mlp2_wp;
    const bool fused2 = packed && (plan & ESMI_FUSE_ATTN_FFN) && enc_attn_ffn_supported(C, n, s->expansion);
    float* x_mid = fused2 ? y1 : x_out;
    // one-kernel-per-op attention with folded weights (esmi.h): the Linear behind the merge convs is x M (h*C wide) instead of qkv
    const bool folded = !fused2 && h >= 2 && w->qk_w && w->qk_wp && w->vo_w && w->vo_wp;   // (one head: measured no gain per op; HISTORY.md 3.3)
    const int nq = folded ? h * C : 3 * h * C;
    EncMergeP m;
    memset(&m, 0, sizeof m);
    m.ids = ids; m.table = embed; m.vocab = s->vocab; m.x_in = ids ? nullptr : x_in;
    m.B = B; m.n_in = s->n_in; m.n_out = n; m.k = s->kernel; m.stride = s->stride; m.pad = s->kernel / 2; m.h = h;
    m.merge_w = w->merge_cwp; m.qkv_w = folded ? w->qk_wp : w->qkv_wp; m.x_out = x_mid; m.qkv = qkv;
    m.nq_override = folded ? nq : 0;
    m.emb_conv = ids ? w->emb_conv : nullptr;
    m.tiles_per_b = (n + 31) / 32;
    EncAttnFfnP f;
    memset(&f, 0, sizeof f);
    f.x = x_mid; f.qkv = qkv; f.B = B; f.N = n; f.C = C; f.h = h; f.scale = 1.0f / sqrtf((float)(C / h));
    f.proj_w = w->proj_wp; f.proj_b = w->proj_b; f.ln1_g = w->ln1_g; f.ln1_b = w->ln1_b;
    f.ffn_w = w->ffn_cwp; f.ffn_b = w->ffn_cb; f.ffn_b0 = w->ffn_cb_first; f.ffn_b2 = w->ffn_cb_last;
    f.mlp2_w = w->mlp2_wp; f.mlp2_b = w->mlp2_b; f.ln2_g = w->ln2_g; f.ln2_b = w->ln2_b;
    f.mask = mask; f.out = x_out;
    f.mask_pool = s->mask_pool > 0 ? s->mask_pool : 1; f.mask_len = s->mask_pool > 0 ? s->mask_len : n;
    if (fused2 && (plan & ESMI_FUSE_MERGE_QKV) && (plan & ESMI_FUSE_BLOCK)) {   // the whole block in one launch
        EncAttnFfnP fb = f;
        fb.m = m;
        fb.x = nullptr; fb.qkv = nullptr;
        if (w->qk_wp && w->vo_wp) {   // weight-folded attention inside the whole-block kernels: a third of the q / k / v contraction
            fb.fold = 1; fb.m.qkv_w = w->qk_wp; fb.m.nq_override = 0; fb.proj_w = w->vo_wp;
        }
        rc = (plan & ESMI_FUSE_CHAIN16) ? launch_enc_block16(fb, s->expansion, s->c_in, st) : ESMI_ERR_UNSUPPORTED;
        if (rc == ESMI_ERR_UNSUPPORTED) rc = launch_enc_block(fb, s->expansion, s->c_in, plan, st);
        if (rc != ESMI_ERR_UNSUPPORTED) return rc;
    }
    if ((plan & ESMI_FUSE_CHAIN16) && folded && !ids && x_in && s->c_in == 128 && C == 256 && s->stride == 2 && n <= 128 && w->merge_cwp) {
        // round 6: base ES's block 1 front (enc_merge256.h) -- the strided merge convolution and the folded attention's query GEMM in ONE launch
        MergeQ256P q;
        q.x_in = x_in; q.x_out = x_mid; q.q = qkv; q.merge_w = w->merge_cwp; q.q_w = w->qk_wp; q.B = B; q.n_in = s->n_in; q.n_out = n;
        q.kernel = s->kernel; q.heads = h;
        rc = launch_enc_merge_q256(q, st);
        if (rc == ESMI_OK) fused1 = true;
        else if (rc != ESMI_ERR_UNSUPPORTED) return rc;
    }
    if (!fused1 && packed && (plan & ESMI_FUSE_MERGE_QKV)) {   // E1: merge conv + 1x1 + qkv as one wave-chain kernel
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
        if (ESMI_CHAIN_SPLIT && w->merge_cwp && (s->c_in & 31) == 0) {   // (the exact-fp32 build's GEMMs read fp32 weights)
            // both merge convolutions as ONE launch on the composed, pre-split weights the chain kernels use (merge_cwp: k x k conv . 1x1,
            // esmi_compose_merge_f32 -> esmi_pack_bfrag_f32): no t_merge round trip, no weight split per wave (round 5)
            p.c_out = C; p.W = nullptr; p.Wp = w->merge_cwp; p.out = x_mid; p.ldo = C;
            if ((rc = launch_convgemm(p, st))) return rc;
        } else {
            p.W = w->merge_w; p.out = t_merge; p.ldo = s->c_in;
            if ((rc = launch_convgemm(p, st))) return rc;
            // merge 1x1, networks.py:67
            p = conv_defaults();
            p.B = B; p.n_in = n; p.c_in = s->c_in; p.n_out = n; p.c_out = C;
            p.A = t_merge; p.lda = s->c_in; p.W = w->merge1_w; p.out = x_mid; p.ldo = C;
            if ((rc = launch_convgemm(p, st))) return rc;
        }
        // qkv Linear (bias-free), blocks.py:44
        p = conv_defaults();
        p.B = B; p.n_in = n; p.c_in = C; p.n_out = n; p.c_out = nq;
        p.A = x_mid; p.lda = C; p.W = folded ? w->qk_w : w->qkv_w; p.Wp = folded ? w->qk_wp : w->qkv_wp; p.out = qkv; p.ldo = nq;
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
    AttnP a = {};
    a.B = B; a.N = n; a.C = C; a.h = h; a.ctx = ctx;
    a.scale = 1.0f / sqrtf((float)(C / h));
    if (folded) {   // q_h = x M_h (the `qkv` buffer, h*C wide); keys = values = x, shared by the heads
        a.q = qkv; a.ldq = h * C; a.hsq = C;
        a.k = a.v = x_out; a.ldk = a.ldv = C; a.hsk = a.hsv = 0;
    } else {
        a.qkv = qkv;
    }
    if ((rc = launch_attn(a, st))) return rc;
    if ((plan & ESMI_FUSE_CHAIN16) && packed && !folded && C == 64 && h == 1 && s->expansion == 1 && n <= 256) {
        // round 6: everything behind the attention in ONE launch (enc_ffn64.h) instead of three GEMM launches through HBM
        PostAttn64P q;
        q.ctx = ctx; q.x = x_out; q.out = x_out; q.proj_w = w->proj_wp; q.ffn_w = w->ffn_cwp; q.mlp2_w = w->mlp2_wp;
        q.proj_b = w->proj_b; q.ln1_g = w->ln1_g; q.ln1_b = w->ln1_b; q.ffn_b = w->ffn_cb; q.ffn_b0 = w->ffn_cb_first; q.ffn_b2 = w->ffn_cb_last;
        q.mlp2_b = w->mlp2_b; q.ln2_g = w->ln2_g; q.ln2_b = w->ln2_b; q.rowmask = mask; q.B = B; q.N = n;
        rc = launch_enc_post_attn64(q, st);
        if (rc != ESMI_ERR_UNSUPPORTED) return rc;
    }
    if ((plan & ESMI_FUSE_CHAIN16) && packed && folded && ffn_folded && C == 128 && h == 2 && s->expansion == 2 && n <= 256) {
        // round 6: the same for base ES's block 0 (enc_ffn128.h): ctx = the two heads' P_h x, the projection is the folded [Wv_h^T Wp_h^T]
        PostAttn128P q;
        q.ctx = ctx; q.x = x_out; q.y1 = y1; q.out = x_out; q.proj_w = w->vo_wp; q.ffn_w = w->ffn_cwp; q.mlp2_w = w->mlp2_wp;
        q.proj_b = w->proj_b; q.ln1_g = w->ln1_g; q.ln1_b = w->ln1_b; q.ffn_b = w->ffn_cb; q.ffn_b0 = w->ffn_cb_first; q.ffn_b2 = w->ffn_cb_last;
        q.mlp2_b = w->mlp2_b; q.ln2_g = w->ln2_g; q.ln2_b = w->ln2_b; q.rowmask = mask; q.B = B; q.N = n;
        rc = launch_enc_post_attn128(q, st);
        if (rc != ESMI_ERR_UNSUPPORTED) return rc;
    }
    // proj + residual + LN1 + mask, blocks.py:65 + networks.py:73-75  (folded: ctx holds P_h x, the matrix is [O_h])
    p = conv_defaults();
    p.B = B; p.n_in = n; p.c_in = h * C; p.n_out = n; p.c_out = C;
    p.A = ctx; p.lda = h * C; p.W = folded ? w->vo_w : w->proj_w; p.Wp = folded ? w->vo_wp : w->proj_wp; p.bias = w->proj_b;
    p.res = x_out; p.ldr = C; p.ln_g = w->ln1_g; p.ln_b = w->ln1_b; p.rowmask = mask;
    p.out = y1; p.ldo = C;
    if ((rc = launch_convgemm(p, st))) return rc;
    // MixFFN, blocks.py:22-29
    if (ffn_folded) {   // Linear folded into the k = 3 conv (esmi.h, ffn_cw): one contraction C -> E, position-dependent bias at the two ends
        p = conv_defaults();
        p.B = B; p.n_in = n; p.c_in = C; p.n_out = n; p.c_out = E; p.k = 3; p.pad = 1;
        p.A = y1; p.lda = C; p.W = w->ffn_cw; p.Wp = w->ffn_cwp; p.bias = w->ffn_cb; p.bias_first = w->ffn_cb_first; p.bias_last = w->ffn_cb_last;
        p.act = ACT_GELU; p.out = m2; p.ldo = E;
        if ((rc = launch_convgemm(p, st))) return rc;
    } else {
        p = conv_defaults();
        p.B = B; p.n_in = n; p.c_in = C; p.n_out = n; p.c_out = E;
        p.A = y1; p.lda = C; p.W = w->mlp1_w; p.Wp = w->mlp1_wp; p.bias = w->mlp1_b; p.out = m1; p.ldo = E;
        if ((rc = launch_convgemm(p, st))) return rc;
        p = conv_defaults();
        p.B = B; p.n_in = n; p.c_in = E; p.n_out = n; p.c_out = E; p.k = 3; p.pad = 1;
        p.A = m1; p.lda = E; p.W = w->conv_w; p.Wp = w->conv_wp; p.bias = w->conv_b; p.act = ACT_GELU; p.out = m2; p.ldo = E;
        if ((rc = launch_convgemm(p, st))) return rc;
    }
    // mlp2 + residual + LN2 + mask, networks.py:80-83
    p = conv_defaults();
    p.B = B; p.n_in = n; p.c_in = E; p.n_out = n; p.c_out = C;
    p.A = m2; p.lda = E; p.W = w->mlp2_w; p.Wp = w->mlp2_wp; p.bias = w->mlp2_b;
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
        p.A = feats[i]; p.lda = ci; p.W = w->mlp_w[i]; p.Wp = w->mlp_wp[i]; p.bias = w->mlp_b[i];
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
            p.A = tmp; p.lda = dim; p.W = w->up_w[i]; p.Wp = w->up_wp[i]; p.bias = w->up_b[i];   // (the streaming kernel reads the pre-split blob: round 5)
            p.out = cat; p.ldo = dim * depth; p.o_coff = i * dim;
            if ((rc = launch_convgemm(p, st))) return rc;
        }
    }
    ConvGemmP p = conv_defaults();  // Linear(depth*dim, dim) + masked_fill, networks.py:215-217
    p.B = B; p.n_in = T; p.c_in = dim * depth; p.n_out = T; p.c_out = dim;
    p.A = cat; p.lda = dim * depth; p.W = w->fuse_w; p.Wp = w->fuse_wp; p.bias = w->fuse_b; p.rowmask = mask;
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
        p.A = feat; p.lda = 4 * dim; p.a_coff = 0; p.W = pw[q]->conv1_w; p.Wp = pw[q]->conv1_wp; p.bias = pw[q]->conv1_b; p.act = ACT_RELU;
        p.ln_g = pw[q]->ln1_g; p.ln_b = pw[q]->ln1_b; p.post_relu = 1; p.out = t1; p.ldo = dim;
        if ((rc = launch_convgemm(p, st))) return rc;
        // conv2 + ReLU; pred = Linear(dim,1) on the PRE-norm2 tensor (:157-160); duration: ReLU + features = LN2
        p = conv_defaults();
        p.B = B; p.n_in = T; p.c_in = dim; p.n_out = T; p.c_out = dim; p.k = 3; p.pad = 1;
        p.A = t1; p.lda = dim; p.W = pw[q]->conv2_w; p.Wp = pw[q]->conv2_wp; p.bias = pw[q]->conv2_b; p.act = ACT_RELU;
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
    if (dim & 3) return ESMI_ERR_UNSUPPORTED;                  // (dim = embed_dim // reduction: 32 / 64 / 128 for the published sizes)
    const long n = (long)B * T * (dim >> 2);                   // one thread per four channels
    ESMI_LAUNCH(va_tail_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, v);
    return launch_status();
}

size_t esmi_fuse_variance_adaptor_workspace_bytes(int B, int T, int dim, int depth) {
    return esmi_fuse_workspace_bytes(B, T, dim, depth) + esmi_variance_adaptor_workspace_bytes(B, T, dim);
}

// the chain kernels' parameter block of the fused Fuse + variance adaptor stage (enc_fuse_va_kernel / enc_va16_kernel)
static void fuse_va_chain_params(const esmi_fuse_weights* fw, int depth, int dim, int kernel, int B, int T, const float* const* feats,
                                 const int* n_i, const esmi_predictor_weights* pitch, const esmi_predictor_weights* energy,
                                 const esmi_predictor_weights* duration, const uint8_t* mask, const float* pitch_target,
                                 const float* energy_target, const int32_t* duration_target, float* feat, float* pitch_pred,
                                 float* energy_pred, float* duration_pred, int32_t* pitch_idx, int32_t* energy_idx, int32_t* dur,
                                 int32_t* cum, int32_t* mel_len, const esmi_decoder_head* head_in_chain, float* h0, FuseVaP* out, int* nw) {
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
    p.mask = mask; p.pitch_t = pitch_target; p.energy_t = energy_target; p.dur_t = duration_target;
    p.feat = feat; p.preds[0] = pitch_pred; p.preds[1] = energy_pred; p.preds[2] = duration_pred;
    p.pitch_idx = pitch_idx; p.energy_idx = energy_idx; p.dur = dur;
    fuse_va_plan(T, dim, depth, nw, &p.wgs_per_b, &p.useful, &p.halo);
    const bool scan_fused = cum && p.halo == 0;   // one workgroup sees every duration of its utterance
    p.cum = scan_fused ? cum : nullptr; p.mel_len = scan_fused ? mel_len : nullptr;
    if (head_in_chain) { p.head_w = head_in_chain->proj_wp; p.head_b = head_in_chain->proj_b; p.head_g = head_in_chain->ln_g; p.head_beta = head_in_chain->ln_b; p.h0 = h0; }
    *out = p;
}

// `lean`: the caller (the one-call inference forward) only consumes duration_pred / dur / cum / mel_len / h0 -- when the round-5 chain
// kernel serves the shape and produces h0, the phoneme-rate feature tensor, the pitch / energy predictions and the bucket indices are
// not written at all (16.8 MB of stores per tiny-ES batch that nobody reads: the decoder gathers h0)
static int fuse_variance_adaptor(const esmi_fuse_weights* fw, int depth, int dim, int kernel, int B, int T,
                                 const float* const* feats, const int* n_i, const esmi_predictor_weights* pitch,
                                 const esmi_predictor_weights* energy, const esmi_predictor_weights* duration,
                                 const uint8_t* mask, const float* pitch_target, const float* energy_target,
                                 const int32_t* duration_target, float* feat, float* pitch_pred, float* energy_pred,
                                 float* duration_pred, int32_t* pitch_idx, int32_t* energy_idx, int32_t* dur,
                                 int32_t* cum, int32_t* mel_len, const esmi_decoder_head* head, float* h0, int plan,
                                 void* workspace, size_t workspace_bytes, esmi_stream_t stream, bool lean);
int esmi_fuse_variance_adaptor_f32(const esmi_fuse_weights* fw, int depth, int dim, int kernel, int B, int T,
                                   const float* const* feats, const int* n_i, const esmi_predictor_weights* pitch,
                                   const esmi_predictor_weights* energy, const esmi_predictor_weights* duration,
                                   const uint8_t* mask, const float* pitch_target, const float* energy_target,
                                   const int32_t* duration_target, float* feat, float* pitch_pred, float* energy_pred,
                                   float* duration_pred, int32_t* pitch_idx, int32_t* energy_idx, int32_t* dur,
                                   int32_t* cum, int32_t* mel_len, const esmi_decoder_head* head, float* h0, int plan,
                                   void* workspace, size_t workspace_bytes, esmi_stream_t stream) {
    return fuse_variance_adaptor(fw, depth, dim, kernel, B, T, feats, n_i, pitch, energy, duration, mask, pitch_target, energy_target,
                                 duration_target, feat, pitch_pred, energy_pred, duration_pred, pitch_idx, energy_idx, dur, cum, mel_len,
                                 head, h0, plan, workspace, workspace_bytes, stream, false);
}
static int fuse_variance_adaptor(const esmi_fuse_weights* fw, int depth, int dim, int kernel, int B, int T,
                                 const float* const* feats, const int* n_i, const esmi_predictor_weights* pitch,
                                 const esmi_predictor_weights* energy, const esmi_predictor_weights* duration,
                                 const uint8_t* mask, const float* pitch_target, const float* energy_target,
                                 const int32_t* duration_target, float* feat, float* pitch_pred, float* energy_pred,
                                 float* duration_pred, int32_t* pitch_idx, int32_t* energy_idx, int32_t* dur,
                                 int32_t* cum, int32_t* mel_len, const esmi_decoder_head* head, float* h0, int plan,
                                 void* workspace, size_t workspace_bytes, esmi_stream_t stream, bool lean) {
    if ((cum == nullptr) != (mel_len == nullptr)) return ESMI_ERR_ARG;
    if (h0 && (!head || (!head->proj_wp && !head->proj_w) || !head->proj_b || !head->ln_g || !head->ln_b)) return ESMI_ERR_ARG;
    if (!fw || !feats || !n_i || !pitch || !energy || !duration || !feat || !pitch_pred || !energy_pred ||
        !duration_pred || !dur || depth < 1 || depth > ESMI_MAX_DEPTH)
        return ESMI_ERR_ARG;
    const bool chain = fuse_va_chain_ok(fw, depth, dim, kernel, n_i[0], T, pitch, energy, duration, plan);
    const bool head_in_chain = h0 && fuse_va_head_ok(chain, dim, head);
    if (h0 && !head_in_chain && !(head_gemm_ok(head) && head->d4 == 4 * dim)) return ESMI_ERR_UNSUPPORTED;
    for (int i = 1; i < depth && chain; ++i)
        if ((n_i[i] - 1) * (1 << i) + kernel < T) return ESMI_ERR_UNSUPPORTED;   // torch.cat would raise in the reference
    if (chain) {
        FuseVaP p;
        int nw;
        if (!pitch->bins || !pitch->emb || !energy->bins || !energy->emb) return ESMI_ERR_ARG;
        fuse_va_chain_params(fw, depth, dim, kernel, B, T, feats, n_i, pitch, energy, duration, mask, pitch_target, energy_target,
                             duration_target, feat, pitch_pred, energy_pred, duration_pred, pitch_idx, energy_idx, dur, cum, mel_len,
                             head_in_chain ? head : nullptr, head_in_chain ? h0 : nullptr, &p, &nw);
        bool scan_fused = cum && p.halo == 0;   // one workgroup sees every duration of its utterance
        int rc16 = ESMI_ERR_UNSUPPORTED;
        if ((plan & ESMI_FUSE_CHAIN16) && scan_fused == (cum != nullptr)) {
            FuseVaP q = p;
            if (lean && head_in_chain) { q.feat = nullptr; q.preds[0] = q.preds[1] = nullptr; q.pitch_idx = q.energy_idx = nullptr; }
            rc16 = launch_enc_va16(q, dim, kernel, S(stream));
        }
        bool head_done = head_in_chain;
        if (rc16 == ESMI_ERR_UNSUPPORTED && (plan & ESMI_FUSE_CHAIN16) && !head_in_chain) {
            // dim = 64, T <= 256 (round 6): one workgroup per utterance, so the length regulator's scan runs inside it -- and the decoder's
            // phoneme-rate first stage too when the caller wants h0 (4 dim = dx2 = 256, pre-split weights); a lean caller then gets
            // neither the feature rows nor the pitch / energy outputs written
            FuseVaP q = p;
            q.cum = cum; q.mel_len = mel_len;
            const bool head64 = h0 && head->proj_wp && head->d4 == 4 * dim && head->dx2 == 4 * dim;
            if (head64) {
                q.head_w = head->proj_wp; q.head_b = head->proj_b; q.head_g = head->ln_g; q.head_beta = head->ln_b; q.h0 = h0;
                if (lean) { q.feat = nullptr; q.preds[0] = q.preds[1] = nullptr; q.pitch_idx = q.energy_idx = nullptr; }
            }
            rc16 = launch_enc_va64(q, dim, kernel, S(stream));
            if (rc16 == ESMI_OK) { scan_fused = cum != nullptr; head_done = head64; }
        }
        if (rc16 == ESMI_ERR_UNSUPPORTED) rc16 = launch_enc_fuse_va(p, dim, kernel, nw, head_in_chain, S(stream));
        if (rc16) return rc16;
        if (cum && !scan_fused) ESMI_LAUNCH(length_regulate_kernel, dim3(B), dim3(64), 0, S(stream), dur, T, cum, mel_len, (int*)nullptr);
        if (h0 && !head_done) return esmi_decoder_head_f32(head, (long)B * T, feat, h0, stream);
        return launch_status();
    }
    if (!workspace || workspace_bytes < esmi_fuse_variance_adaptor_workspace_bytes(B, T, dim, depth)) return ESMI_ERR_WORKSPACE;
    const size_t fws = esmi_fuse_workspace_bytes(B, T, dim, depth);
    // dim = 128, two levels, T <= 256 (round 6): the Fuse stage as ONE launch (enc_fuse128.h) instead of four GEMM launches through HBM
    int rc = ESMI_ERR_UNSUPPORTED;
    if ((plan & ESMI_FUSE_CHAIN16) && dim == 128 && depth == 2 && fw->mlp_wp[0] && fw->mlp_wp[1] && fw->up_wp[1] && fw->fuse_wp) {
        FuseVaP q;
        memset(&q, 0, sizeof q);
        q.B = B; q.T = T; q.depth = depth; q.kernel = kernel;
        for (int i = 0; i < depth; ++i) {
            q.feats[i] = feats[i]; q.n_i[i] = n_i[i];
            q.mlp_w[i] = fw->mlp_wp[i]; q.mlp_b[i] = fw->mlp_b[i]; q.up_w[i] = fw->up_wp[i]; q.up_b[i] = fw->up_b[i];
        }
        q.fuse_w = fw->fuse_wp; q.fuse_b = fw->fuse_b; q.mask = mask; q.feat = feat;
        rc = launch_enc_fuse128(q, dim, kernel, S(stream));
    }
    if (rc == ESMI_ERR_UNSUPPORTED) rc = esmi_fuse_f32(fw, depth, dim, kernel, B, T, feats, n_i, mask, feat, 4 * dim, workspace, fws, stream);
    if (rc) return rc;
    // dim = 128, T <= 256 (round 6): the three predictors as ONE launch -- a workgroup per (utterance, predictor) with the hidden rows in
    // registers, bucketize / embeddings / duration features / rounding and the length regulator's scan inside (enc_pred128.h) -- instead
    // of six GEMM launches through HBM + va_tail_kernel + length_regulate_kernel
    rc = ESMI_ERR_UNSUPPORTED;
    if ((plan & ESMI_FUSE_CHAIN16) && dim == 128 && pitch->conv1_wp && pitch->conv2_wp && energy->conv1_wp && energy->conv2_wp &&
        duration->conv1_wp && duration->conv2_wp) {
        Pred128P q;
        memset(&q, 0, sizeof q);
        const esmi_predictor_weights* pw[3] = {pitch, energy, duration};
        for (int k = 0; k < 3; ++k) {
            PredW& d = q.pred[k];
            d.conv1_w = pw[k]->conv1_wp; d.conv1_b = pw[k]->conv1_b; d.ln1_g = pw[k]->ln1_g; d.ln1_b = pw[k]->ln1_b;
            d.conv2_w = pw[k]->conv2_wp; d.conv2_b = pw[k]->conv2_b; d.ln2_g = pw[k]->ln2_g; d.ln2_b = pw[k]->ln2_b;
            d.lin_w = pw[k]->lin_w; d.lin_b = pw[k]->lin_b; d.bins = pw[k]->bins; d.emb = pw[k]->emb;
        }
        q.mask = mask; q.pitch_t = pitch_target; q.energy_t = energy_target; q.dur_t = duration_target;
        q.feat = feat; q.preds[0] = pitch_pred; q.preds[1] = energy_pred; q.preds[2] = duration_pred;
        q.pitch_idx = pitch_idx; q.energy_idx = energy_idx; q.dur = dur; q.cum = cum; q.mel_len = mel_len; q.B = B; q.T = T;
        rc = launch_enc_pred128(q, dim, S(stream));
    }
    if (rc == ESMI_ERR_UNSUPPORTED) {
        rc = esmi_variance_adaptor_f32(pitch, energy, duration, dim, B, T, mask, pitch_target, energy_target,
                                       duration_target, feat, pitch_pred, energy_pred, duration_pred, pitch_idx, energy_idx,
                                       dur, static_cast<char*>(workspace) + fws, workspace_bytes - fws, stream);
        if (rc) return rc;
        if (cum) ESMI_LAUNCH(length_regulate_kernel, dim3(B), dim3(64), 0, S(stream), dur, T, cum, mel_len, (int*)nullptr);
    } else if (rc) {
        return rc;
    }
    if (h0) return esmi_decoder_head_f32(head, (long)B * T, feat, h0, stream);
    return launch_status();
}

// MelDecoder's first stage at phoneme rate as one launch: GEMM (k = 1) + bias + tanh + LayerNorm in the epilogue (networks.py:291-293)
int esmi_decoder_head_f32(const esmi_decoder_head* head, long rows, const float* feat, float* h0, esmi_stream_t stream) {
    if (!head_gemm_ok(head) || !feat || !h0 || rows <= 0 || rows > 0x7fffffffL) return head_gemm_ok(head) ? ESMI_ERR_ARG : ESMI_ERR_UNSUPPORTED;
    ConvGemmP p = conv_defaults();
    p.B = 1; p.n_in = p.n_out = (int)rows; p.c_in = head->d4; p.c_out = head->dx2;
    p.A = feat; p.lda = head->d4; p.W = head->proj_w; p.bias = head->proj_b; p.act = ACT_TANH;
    p.ln_g = head->ln_g; p.ln_b = head->ln_b; p.out = h0; p.ldo = head->dx2;
    p.Wp = head->proj_wp;            // (optional) pre-split fragments: the LDS-staged kernel then brings the weight tiles in by LDS-DMA
    return launch_convgemm(p, S(stream));
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
    AttnP a = {};                     // softmax(q k^T scale) v, blocks.py:49-64 (scores not masked)
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
    p.A = x; p.lda = ldx; p.W = w->conv1_w; p.Wp = w->conv1_wp; p.bias = w->conv1_b; p.act = ACT_RELU;
    p.ln_g = w->ln1_g; p.ln_b = w->ln1_b; p.post_relu = 1; p.out = t1; p.ldo = dim;
    int rc = launch_convgemm(p, st);
    if (rc) return rc;
    p = conv_defaults();              // conv2 + ReLU; y = Linear(dim,1) on the PRE-norm2 tensor (:157-160); duration: ReLU + features = LN2
    p.B = B; p.n_in = T; p.c_in = dim; p.n_out = T; p.c_out = dim; p.k = 3; p.pad = 1;
    p.A = t1; p.lda = dim; p.W = w->conv2_w; p.Wp = w->conv2_wp; p.bias = w->conv2_b; p.act = ACT_RELU;
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
    o->h0 = off; off += (a->head.proj_wp || a->head.proj_w) ? align256(rows * a->head.dx2 * 4) : 0;
    o->total = off;
    return ESMI_OK;
}
}  // namespace

// The decoder re-uses the encoder side's scratch for its carried rows (dx2 = 256 chunk walk).  When that scratch is too small for a
// batch (few utterances, long output) the arena carries a tail region for them behind everything else -- the layout of the other
// regions never depends on L_out, so a stage-2 call with the length filled in later sees the same offsets.
static size_t fwd_dec_tail(const esmi_forward_args* a, const FwdArena& o) {
    // (L_out unknown at sizing time -- the reference's own call style: stage 1, host sync, stage 2 -- : the need of an arbitrarily long
    // output, which is bounded: the segmentation never uses more workgroups than max(B, one per CU).  Both call styles then run the same
    // decoder form)
    const size_t need = esmi_mel_decoder_workspace_bytes(&a->dec_shape, a->B, a->L_out > 0 ? a->L_out : -1);
    return need > o.feat - o.ws ? align256(need) : 0;
}
size_t esmi_forward_arena_bytes(const esmi_forward_args* a) {
    FwdArena o;
    return fwd_arena(a, &o) == ESMI_OK ? o.total + fwd_dec_tail(a, o) : 0;
}

static int forward_impl(const esmi_forward_args* a, int stage, esmi_stream_t stream);
int esmi_phoneme2mel_forward_f32(const esmi_forward_args* a, int stage, esmi_stream_t stream) {
    if (!a) return ESMI_ERR_ARG;
    if (!a->range_flag) return forward_impl(a, stage, stream);
#if ESMI_RANGE_CHECK
    // validation mode: clear the word, point every translation unit's kernels at it, run, wait, read it back
    hipError_t e = hipMemsetAsync(a->range_flag, 0, sizeof(int32_t), S(stream));
    if (e != hipSuccess) return (int)e;
    int (*const setters[])(int*) = {set_range_flag_abi, set_range_flag_convgemm, set_range_flag_attention, set_range_flag_enc_merge,
                                    set_range_flag_enc_block, set_range_flag_enc_attn_ffn, set_range_flag_enc_fuse_va, set_range_flag_enc_va16, set_range_flag_enc_va64, set_range_flag_enc_pred128, set_range_flag_enc_block16,
                                    set_range_flag_decoder, set_range_flag_dec_128_5, set_range_flag_dec_128_3, set_range_flag_dec_256_5,
                                    set_range_flag_dec_256_3, set_range_flag_hifigan, set_range_flag_train};
    for (auto set : setters)
        if (int rc = set(reinterpret_cast<int*>(a->range_flag))) return rc;
    int rc = forward_impl(a, stage, stream);
    int32_t flag = 0;
    if (!rc) rc = read_device_flag(a->range_flag, S(stream), &flag);
    for (auto set : setters) set(nullptr);
    return rc ? rc : (flag ? ESMI_ERR_RANGE : ESMI_OK);
#else
    return ESMI_ERR_UNSUPPORTED;   // this library was built without the range check (libesmi_checked.so has it)
#endif
}
static int forward_impl(const esmi_forward_args* a, int stage, esmi_stream_t stream) {
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
    float* h0 = (a->head.proj_wp || a->head.proj_w) ? F(o.h0) : nullptr;
    const uint8_t* mask = a->mask;
    bool enc_done = false;
    if (stage != 2 && (plan & ESMI_FUSE_ALL) == ESMI_FUSE_ALL && a->depth == 2 && T <= 128) {
        // ---- the whole encoder side as ONE launch (round 5: enc_all16_kernel = block 0 | block 1 | Fuse + variance adaptor + head behind
        // each other in one workgroup per utterance) when all three chain16 kernels serve their shapes
        esmi_encoder_block_shape sh[2];
        // (the chain16 bodies are built for MixFFN expansion 1 only, like launch_enc_block16: anything else takes the per-block path below)
        bool ok = a->shapes[0].expansion == 1 && a->shapes[1].expansion == 1;
        int n_in = T;
        for (int i = 0; i < 2 && ok; ++i) {
            sh[i] = a->shapes[i];
            sh[i].B = B; sh[i].n_in = n_in; sh[i].plan = plan; sh[i].mask_pool = 1; sh[i].mask_len = T;
            if (mask) {
                sh[i].mask_pool = (int)nearbyint((double)T / o.n[i]);
                ok = (T + sh[i].mask_pool - 1) / sh[i].mask_pool == o.n[i];
            }
            n_in = o.n[i];
        }
        EncAttnFfnP b0, b1;
        ok = ok && block_chain_params(&a->blocks[0], &sh[0], a->ids, a->embed, nullptr, mask, F(o.feats[0]), &b0) &&
             block_chain_params(&a->blocks[1], &sh[1], nullptr, nullptr, F(o.feats[0]), mask, F(o.feats[1]), &b1);
        const bool chain = fuse_va_chain_ok(&a->fuse, a->depth, a->dim, a->fuse_kernel, o.n[0], T, &a->pitch, &a->energy, &a->duration, plan);
        const bool want_head = a->head.proj_wp || a->head.proj_w;
        const bool head_in_chain = want_head && fuse_va_head_ok(chain, a->dim, &a->head);
        ok = ok && chain && (!want_head || head_in_chain) && a->pitch.bins && a->pitch.emb && a->energy.bins && a->energy.emb &&
             (o.n[1] - 1) * 2 + a->fuse_kernel >= T;
        if (ok) {
            const float* feats[2] = {F(o.feats[0]), F(o.feats[1])};
            const bool lean = !a->pitch_pred && !a->energy_pred && !a->pitch_idx && !a->energy_idx && head_in_chain;
            FuseVaP va;
            int nw_unused;
            fuse_va_chain_params(&a->fuse, 2, a->dim, a->fuse_kernel, B, T, feats, o.n, &a->pitch, &a->energy, &a->duration, mask, nullptr, nullptr,
                                 a->dur_forced, lean ? nullptr : feat, lean ? nullptr : (a->pitch_pred ? a->pitch_pred : F(o.preds[0])),
                                 lean ? nullptr : (a->energy_pred ? a->energy_pred : F(o.preds[1])), a->duration_pred,
                                 lean ? nullptr : (a->pitch_idx ? a->pitch_idx : I(o.idx[0])), lean ? nullptr : (a->energy_idx ? a->energy_idx : I(o.idx[1])),
                                 a->dur ? a->dur : I(o.dur), cum, a->mel_len, head_in_chain ? &a->head : nullptr, head_in_chain ? h0 : nullptr, &va,
                                 &nw_unused);
            rc = va.cum ? launch_enc_all16(b0, b1, sh[1].c_in, va, a->dim, a->fuse_kernel, S(stream)) : ESMI_ERR_UNSUPPORTED;
            if (rc == ESMI_OK) enc_done = true;
            else if (rc != ESMI_ERR_UNSUPPORTED) return rc;
        }
        if (enc_done && a->lmax_dev && (rc = esmi_max_i32(a->mel_len, B, a->lmax_dev, stream))) return rc;
    }
    if (stage != 2 && !enc_done) {
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
                                                              &a->duration, plan), a->dim, &a->head) ||
                             (head_gemm_ok(&a->head) && a->head.d4 == 4 * a->dim && a->head.dx2 == a->dec_shape.dx2);
        // (lean: nobody reads the arena's feature / prediction / index buffers when the caller did not ask for them and h0 is produced)
        const bool lean = !a->pitch_pred && !a->energy_pred && !a->pitch_idx && !a->energy_idx;
        rc = fuse_variance_adaptor(&a->fuse, a->depth, a->dim, a->fuse_kernel, B, T, feats, o.n, &a->pitch, &a->energy,
                                   &a->duration, mask, nullptr, nullptr, a->dur_forced, feat, pp, ep, a->duration_pred,
                                   pi, ei, dur, cum, a->mel_len, head_ok ? &a->head : nullptr, head_ok ? h0 : nullptr, plan,
                                   base + o.ws, wsb, stream, lean);
        if (rc) return rc;
        if (a->lmax_dev && (rc = esmi_max_i32(a->mel_len, B, a->lmax_dev, stream))) return rc;
    }
    if (stage != 1) {
        if (a->L_out <= 0) return ESMI_OK;   // every duration zero: empty mel, nothing to launch
        if (!a->mel || !a->dec_blob) return ESMI_ERR_ARG;
        // (a stage-2 call re-derives whether stage 1 produced the phoneme-rate head: the same static test)
        const bool head_ok = fuse_va_head_ok(fuse_va_chain_ok(&a->fuse, a->depth, a->dim, a->fuse_kernel, o.n[0], T, &a->pitch, &a->energy,
                                                              &a->duration, plan), a->dim, &a->head) ||
                             (head_gemm_ok(&a->head) && a->head.d4 == 4 * a->dim && a->head.dx2 == a->dec_shape.dx2);
        // the decoder's carried rows: the encoder side's scratch (free again), or the arena's tail region when that is too small
        const size_t dec_need = esmi_mel_decoder_workspace_bytes(&a->dec_shape, B, a->L_out);
        const bool use_tail = dec_need > o.feat - o.ws && a->arena_bytes >= o.total + dec_need;
        rc = esmi_mel_decoder_f32(a->dec_blob, &a->dec_shape, feat, head_ok ? h0 : nullptr, cum, a->mel_len,
                                  a->lmax_host < 0 ? a->lmax_dev : nullptr, a->lmax_host, mask != nullptr && B > 1, B, T, a->L_out,
                                  a->mel, use_tail ? base + o.total : base + o.ws, use_tail ? a->arena_bytes - o.total : o.feat - o.ws,
                                  stream);
        if (rc) return rc;
    }
    return ESMI_OK;
}

}  // extern "C"
