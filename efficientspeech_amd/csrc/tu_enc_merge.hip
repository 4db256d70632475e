// esmi C-ABI, translation unit "tu_enc_merge.hip": merge conv + 1x1 + qkv chain kernel (enc_merge_qkv.h)
// One of several translation units of libesmi.so (compiled in parallel by __graft_entry__.build(); the simulator build
// tools/wavesim/build.sh compiles the same files with the host compiler).  Internal launchers are declared in launch.h.
#include "launch.h"

using namespace esmi;
ESMI_TU_RANGE_SETTER(enc_merge)
ESMI_TU_CHAIN_TRACE_SETTER(enc_merge)

namespace esmi {

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

}  // namespace esmi
