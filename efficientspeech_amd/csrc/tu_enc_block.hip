// esmi C-ABI, translation unit "tu_enc_block.hip": whole encoder block in one launch (enc_attn_ffn.h, NCI > 0 instantiations)
// One of several translation units of libesmi.so (compiled in parallel by __graft_entry__.build(); the simulator build
// tools/wavesim/build.sh compiles the same files with the host compiler).  Internal launchers are declared in launch.h.
#include "launch.h"

using namespace esmi;
ESMI_TU_RANGE_SETTER(enc_block)
ESMI_TU_CHAIN_TRACE_SETTER(enc_block)

namespace esmi {

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


}  // namespace esmi
