// esmi C-ABI, translation unit "tu_enc_fuse_va.hip": Fuse + variance-adaptor chain kernel (enc_fuse_va.h)
// One of several translation units of libesmi.so (compiled in parallel by __graft_entry__.build(); the simulator build
// tools/wavesim/build.sh compiles the same files with the host compiler).  Internal launchers are declared in launch.h.
#include "launch.h"

using namespace esmi;
ESMI_TU_RANGE_SETTER(enc_fuse_va)
ESMI_TU_CHAIN_TRACE_SETTER(enc_fuse_va)

namespace esmi {

// Fuse + variance adaptor (+ length-regulator scan, + the decoder head at phoneme rate) in one launch
int launch_enc_fuse_va(const FuseVaP& p, int dim, int kernel, int nw, bool head, hipStream_t st) {
    dim3 grid(p.B * p.wgs_per_b), block(64 * nw);
    const int lds = fuse_va_lds_floats(dim, p.depth, nw, head) * (int)sizeof(float);
    static AttrOnce once[4];
    const void* fns[4] = {reinterpret_cast<const void*>(enc_fuse_va_kernel<1, 3>), reinterpret_cast<const void*>(enc_fuse_va_kernel<2, 3>),
                          reinterpret_cast<const void*>(enc_fuse_va_kernel<1, 5>), reinterpret_cast<const void*>(enc_fuse_va_kernel<2, 5>)};
    for (int q = 0; q < 4; ++q)
        if (int rc = raise_lds_limit(fns[q], once[q])) return rc;
    if (dim == 32 && kernel == 3) ESMI_LAUNCH((enc_fuse_va_kernel<1, 3>), grid, block, lds, st, p);
    else if (dim == 64 && kernel == 3) ESMI_LAUNCH((enc_fuse_va_kernel<2, 3>), grid, block, lds, st, p);
    else if (dim == 32 && kernel == 5) ESMI_LAUNCH((enc_fuse_va_kernel<1, 5>), grid, block, lds, st, p);
    else if (dim == 64 && kernel == 5) ESMI_LAUNCH((enc_fuse_va_kernel<2, 5>), grid, block, lds, st, p);
    else return ESMI_ERR_UNSUPPORTED;
    return launch_status();
}

}  // namespace esmi
