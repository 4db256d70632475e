// esmi C-ABI, translation unit "tu_enc_attn_ffn.hip": attention + proj + LN + MixFFN + LN chain kernel (enc_attn_ffn.h)
// One of several translation units of libesmi.so (compiled in parallel by __graft_entry__.build(); the simulator build
// tools/wavesim/build.sh compiles the same files with the host compiler).  Internal launchers are declared in launch.h.
#include "launch.h"

using namespace esmi;
ESMI_TU_RANGE_SETTER(enc_attn_ffn)
ESMI_TU_CHAIN_TRACE_SETTER(enc_attn_ffn)

namespace esmi {

// E2: attention + proj + LN1 + MixFFN + LN2 in one launch.
int launch_enc_attn_ffn(const EncAttnFfnP& p, int expansion, int plan, hipStream_t st) {
    // N <= 128 only: the 8-key-tile instantiations (N <= 256) spilled 84 .. 820 B per lane at 512 + 256 registers and were slower than the
    // per-op launches (`enc_attn_ffn_supported`, launch.h, never selected them); round 6 stops building them -- a longer sequence gets
    // ESMI_ERR_UNSUPPORTED here and the LDS-staged attention + GEMM launches from the caller
    if ((p.C & 31) || p.N > 128) return ESMI_ERR_UNSUPPORTED;
    const int nc = p.C / 32, nkt = p.N <= 64 ? 2 : 4;
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
#define ESMI_E2K(NC, E) ESMI_E2(2, NC, E) ESMI_E2(4, NC, E)
    ESMI_E2K(1, 1) ESMI_E2K(2, 1) ESMI_E2K(4, 1) ESMI_E2K(4, 2)
#undef ESMI_E2K
#undef ESMI_E2
    return ESMI_ERR_UNSUPPORTED;
}

}  // namespace esmi
