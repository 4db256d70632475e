// esmi C-ABI, translation unit "tu_enc_va16.hip": the round-5 Fuse + variance-adaptor chain kernel (enc_va16.h: 16-row tiles, two
// waves per SIMD, weights once per workgroup through LDS).  Internal launchers are declared in launch.h.
#include "launch.h"
#include "enc_va16.h"

using namespace esmi;
ESMI_TU_RANGE_SETTER(enc_va16)
ESMI_TU_CHAIN_TRACE_SETTER(enc_va16)

namespace esmi {

// dim = 32, two encoder levels, ConvTranspose kernel 3, one workgroup per utterance (T <= 128); ESMI_ERR_UNSUPPORTED otherwise
// (-> enc_fuse_va_kernel).  The split-f16 build only: the exact-fp32 library keeps the round-1 kernel.
bool enc_va16_ok(const FuseVaP& p, int dim, int kernel) {
    return dim == kVa16Dim && p.depth == 2 && kernel == 3 && p.T >= 1 && p.T <= 16 * kVa16MaxWaves && p.n_i[0] == p.T;
}

int launch_enc_va16(const FuseVaP& p, int dim, int kernel, hipStream_t st) {
#if ESMI_CHAIN_SPLIT
    if (!enc_va16_ok(p, dim, kernel)) return ESMI_ERR_UNSUPPORTED;
    const int nw = (p.T + 15) / 16;
    static AttrOnce once;
    if (int rc = raise_lds_limit(reinterpret_cast<const void*>(enc_va16_kernel<3>), once)) return rc;
    ESMI_LAUNCH((enc_va16_kernel<3>), dim3(p.B), dim3(64 * nw), va16_lds_bytes(), st, p);
    return launch_status();
#else
    (void)p; (void)dim; (void)kernel; (void)st;
    return ESMI_ERR_UNSUPPORTED;
#endif
}

}  // namespace esmi
