// esmi C-ABI, translation unit "tu_enc_va64.hip": the round-6 Fuse + variance-adaptor kernel of dim = 64 models (enc_va64.h: activations
// in registers from the input rows to the stored features, LDS for the weights).  Internal launchers are declared in launch.h.
#include "launch.h"
#include "enc_va64.h"
#include "enc_ffn64.h"

using namespace esmi;
ESMI_TU_RANGE_SETTER(enc_va64)

namespace esmi {

// dim = 64, two encoder levels, ConvTranspose kernel 3, one workgroup per utterance (T <= 256); ESMI_ERR_UNSUPPORTED otherwise
// (-> enc_fuse_va_kernel).  The split-f16 build only: the exact-fp32 library keeps the round-1 kernel.
bool enc_va64_ok(const FuseVaP& p, int dim, int kernel) {
    return dim == kVa64Dim && p.depth == 2 && kernel == 3 && p.T >= 1 && p.T <= 32 * kVa64MaxWaves && p.n_i[0] == p.T &&
           (!p.h0 || (p.head_w && p.head_b && p.head_g && p.head_beta));
}

int launch_enc_va64(const FuseVaP& p, int dim, int kernel, hipStream_t st) {
#if ESMI_CHAIN_SPLIT
    if (!enc_va64_ok(p, dim, kernel)) return ESMI_ERR_UNSUPPORTED;
    if (p.T <= 16 * kVa64MaxWaves) {      // one 16-row tile per wave
        static AttrOnce once;
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(enc_va64_kernel<1>), once)) return rc;
        ESMI_LAUNCH((enc_va64_kernel<1>), dim3(p.B), dim3(64 * ((p.T + 15) / 16)), va64_lds_bytes(), st, p);
    } else {                              // two tiles per wave: every weight fragment read from LDS serves 32 rows
        static AttrOnce once;
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(enc_va64_kernel<2>), once)) return rc;
        ESMI_LAUNCH((enc_va64_kernel<2>), dim3(p.B), dim3(64 * ((p.T + 31) / 32)), va64_lds_bytes(), st, p);
    }
    return launch_status();
#else
    (void)p; (void)dim; (void)kernel; (void)st;
    return ESMI_ERR_UNSUPPORTED;
#endif
}

// Everything behind the attention of a C = 64 one-head block (N <= 256) in one launch (enc_ffn64.h); ESMI_ERR_UNSUPPORTED in the exact-fp32 build
int launch_enc_post_attn64(const PostAttn64P& p, hipStream_t st) {
#if ESMI_CHAIN_SPLIT
    if (p.N < 1 || p.N > 32 * kVa64MaxWaves || p.B < 1) return ESMI_ERR_UNSUPPORTED;
    if (p.N <= 16 * kVa64MaxWaves) {
        static AttrOnce once;
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(enc_post_attn64_kernel<1>), once)) return rc;
        ESMI_LAUNCH((enc_post_attn64_kernel<1>), dim3(p.B), dim3(64 * ((p.N + 15) / 16)), ffn64_lds_bytes(), st, p);
    } else {
        static AttrOnce once;
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(enc_post_attn64_kernel<2>), once)) return rc;
        ESMI_LAUNCH((enc_post_attn64_kernel<2>), dim3(p.B), dim3(64 * ((p.N + 31) / 32)), ffn64_lds_bytes(), st, p);
    }
    return launch_status();
#else
    (void)p; (void)st;
    return ESMI_ERR_UNSUPPORTED;
#endif
}

}  // namespace esmi
