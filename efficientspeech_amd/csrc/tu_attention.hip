// esmi C-ABI, translation unit "tu_attention.hip": stand-alone attention kernels (attention.h)
// One of several translation units of libesmi.so (compiled in parallel by __graft_entry__.build(); the simulator build
// tools/wavesim/build.sh compiles the same files with the host compiler).  Internal launchers are declared in launch.h.
#include "launch.h"

using namespace esmi;
ESMI_TU_RANGE_SETTER(attention)

namespace esmi {

// (utterance, head) pairs from which attention is LDS-staged: kAttnLdsMinHeadsDefault (wavesim_shim.h: 128 on the GPU, 1 in the simulator)
int launch_attn(const AttnP& p_in, hipStream_t st) {
    AttnP p = p_in;
    if (!p.q) {                        // the reference's layout: one (B, N, 3, h, C) tensor
        if (!p.qkv) return ESMI_ERR_ARG;
        const int hc = p.h * p.C;
        p.q = p.qkv; p.k = p.qkv + hc; p.v = p.qkv + 2 * hc;
        p.ldq = p.ldk = p.ldv = 3 * hc;
        p.hsq = p.hsk = p.hsv = p.C;
    }
    if (!p.k || !p.v || (p.ldq & 3) || (p.ldk & 3) || (p.ldv & 3) || (p.hsq & 3) || (p.hsk & 3) || (p.hsv & 3)) return ESMI_ERR_ARG;
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
    if (nkt >= 3 && (p.C == 32 || p.C == 64 || p.C % 128 == 0) && attn_lds_bytes(p.N, p.C) <= 150 * 1024 && (long)p.B * p.h >= kAttnLdsMinHeadsDefault) {
        const size_t lds = attn_lds_bytes(p.N, p.C);
        dim3 g2((unsigned)(p.B * p.h));
#define ESMI_ATTN_LDS(NKT, CKV) do { \
            static AttrOnce once; \
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(attn_lds_kernel<NKT, CKV>), once)) return rc; \
            ESMI_LAUNCH((attn_lds_kernel<NKT, CKV>), g2, dim3(64 * NKT), lds, st, p); \
        } while (0)
        const int ck = p.C < 128 ? p.C : 128;
        if (nkt <= 4) {
            if (ck == 32) ESMI_ATTN_LDS(4, 32); else if (ck == 64) ESMI_ATTN_LDS(4, 64); else ESMI_ATTN_LDS(4, 128);
        } else {
            if (ck == 32) ESMI_ATTN_LDS(8, 32); else if (ck == 64) ESMI_ATTN_LDS(8, 64); else ESMI_ATTN_LDS(8, 128);
        }
#undef ESMI_ATTN_LDS
        return launch_status();
    }
#endif
    if (nkt == 1) ESMI_LAUNCH((attn_kernel<1>), grid, block, 0, st, p);
    else if (nkt == 2) ESMI_LAUNCH((attn_kernel<2>), grid, block, 0, st, p);
    else if (nkt <= 4) ESMI_LAUNCH((attn_kernel<4>), grid, block, 0, st, p);
    else ESMI_LAUNCH((attn_kernel<8>), grid, block, 0, st, p);
    return launch_status();
}

}  // namespace esmi
