// esmi C-ABI, translation unit "tu_enc_pred128.hip": the three predictors of a dim = 128 model (base ES), one workgroup per
// (utterance, predictor), with the variance adaptor's tail and the length regulator's scan inside (enc_pred128.h).  Internal
// launchers are declared in launch.h.
#include "launch.h"
#include "enc_pred128.h"
#include "enc_fuse128.h"
#include "enc_ffn128.h"
#include "enc_merge256.h"

using namespace esmi;
ESMI_TU_RANGE_SETTER(enc_pred128)

namespace esmi {

// dim = 128, one workgroup covers the sequence (T <= 256), pre-split conv weights; ESMI_ERR_UNSUPPORTED otherwise (-> the per-op plan).
// The split-f16 build only: the exact-fp32 library keeps the per-op plan.
int launch_enc_pred128(const Pred128P& p, int dim, hipStream_t st) {
#if ESMI_CHAIN_SPLIT
    if (dim != 128 || p.T < 1 || p.T > 32 * kVa64MaxWaves || p.B < 1 || !p.feat || !p.dur || !p.preds[0] || !p.preds[1] || !p.preds[2] ||
        ((p.cum == nullptr) != (p.mel_len == nullptr)))
        return ESMI_ERR_UNSUPPORTED;
    for (int q = 0; q < 3; ++q) {
        const PredW& d = p.pred[q];
        if (!d.conv1_w || !d.conv2_w || !d.conv1_b || !d.conv2_b || !d.ln1_g || !d.ln1_b || !d.lin_w || !d.lin_b) return ESMI_ERR_UNSUPPORTED;
        if (q < 2 && (!d.bins || !d.emb)) return ESMI_ERR_UNSUPPORTED;
        if (q == 2 && (!d.ln2_g || !d.ln2_b)) return ESMI_ERR_UNSUPPORTED;
    }
    static AttrOnce once;
    if (int rc = raise_lds_limit(reinterpret_cast<const void*>(enc_pred128_kernel), once)) return rc;
    ESMI_LAUNCH(enc_pred128_kernel, dim3(p.B, 3), dim3(64 * ((p.T + 31) / 32)), pred128_lds_bytes(), st, p);
    return launch_status();
#else
    (void)p; (void)dim; (void)st;
    return ESMI_ERR_UNSUPPORTED;
#endif
}

// Fuse of a dim = 128 model with two encoder levels as one launch (enc_fuse128.h): feat[:, 0 .. dim) <- the masked fused rows
int launch_enc_fuse128(const FuseVaP& p, int dim, int kernel, hipStream_t st) {
#if ESMI_CHAIN_SPLIT
    if (dim != 128 || p.depth != 2 || (kernel != 3 && kernel != 5) || p.T < 1 || p.T > 32 * kVa64MaxWaves || p.B < 1 || p.n_i[0] != p.T ||
        p.n_i[1] < 1 || p.n_i[1] > (p.T + 1) / 2 || (p.n_i[1] - 1) * 2 + kernel < p.T || !p.feat || !p.feats[0] || !p.feats[1] || !p.mlp_w[0] ||
        !p.mlp_w[1] || !p.up_w[1] || !p.fuse_w || !p.mlp_b[0] || !p.mlp_b[1] || !p.up_b[1] || !p.fuse_b)
        return ESMI_ERR_UNSUPPORTED;
    const dim3 block(64 * ((p.T + 31) / 32));
    if (kernel == 5) {
        static AttrOnce once;
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(enc_fuse128_kernel<5>), once)) return rc;
        ESMI_LAUNCH((enc_fuse128_kernel<5>), dim3(p.B), block, fuse128_lds_bytes(), st, p);
    } else {
        static AttrOnce once;
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(enc_fuse128_kernel<3>), once)) return rc;
        ESMI_LAUNCH((enc_fuse128_kernel<3>), dim3(p.B), block, fuse128_lds_bytes(), st, p);
    }
    return launch_status();
#else
    (void)p; (void)dim; (void)kernel; (void)st;
    return ESMI_ERR_UNSUPPORTED;
#endif
}

// Everything behind the attention of a C = 128, two-head, expansion-2 block (N <= 256) in one launch (enc_ffn128.h)
int launch_enc_post_attn128(const PostAttn128P& p, hipStream_t st) {
#if ESMI_CHAIN_SPLIT
    if (p.N < 1 || p.N > 32 * kVa64MaxWaves || p.B < 1 || !p.ctx || !p.x || !p.y1 || !p.out || !p.proj_w || !p.ffn_w || !p.mlp2_w || !p.proj_b ||
        !p.ln1_g || !p.ln1_b || !p.ffn_b || !p.ffn_b0 || !p.ffn_b2 || !p.mlp2_b || !p.ln2_g || !p.ln2_b)
        return ESMI_ERR_UNSUPPORTED;
    static AttrOnce once;
    if (int rc = raise_lds_limit(reinterpret_cast<const void*>(enc_post_attn128_kernel), once)) return rc;
    ESMI_LAUNCH(enc_post_attn128_kernel, dim3(p.B), dim3(64 * ((p.N + 31) / 32)), ffn128_lds_bytes(), st, p);
    return launch_status();
#else
    (void)p; (void)st;
    return ESMI_ERR_UNSUPPORTED;
#endif
}

// The front of a C = 256 block fed by 128-channel rows (stride 2; N <= 128 output rows) in one launch (enc_merge256.h): base ES's block 1
int launch_enc_merge_q256(const MergeQ256P& p, hipStream_t st) {
#if ESMI_CHAIN_SPLIT
    if (p.kernel != 3 || p.heads != 4 || p.n_out < 1 || p.n_out > 16 * kVa64MaxWaves || p.B < 1 || p.n_out != (p.n_in + 2 * (p.kernel / 2) - p.kernel) / 2 + 1 ||
        !p.x_in || !p.x_out || !p.q || !p.merge_w || !p.q_w)
        return ESMI_ERR_UNSUPPORTED;
    static AttrOnce once;
    if (int rc = raise_lds_limit(reinterpret_cast<const void*>(enc_merge_q256_kernel<3, 4>), once)) return rc;
    ESMI_LAUNCH((enc_merge_q256_kernel<3, 4>), dim3(p.B), dim3(64 * ((p.n_out + 15) / 16)), merge256_lds_bytes(), st, p);
    return launch_status();
#else
    (void)p; (void)st;
    return ESMI_ERR_UNSUPPORTED;
#endif
}

}  // namespace esmi
