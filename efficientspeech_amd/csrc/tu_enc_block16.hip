// esmi C-ABI, translation unit "tu_enc_block16.hip": the round-5 whole-block chain kernels of dim = 32 models (enc_block16.h: 16-row
// tiles, two waves per SIMD, weights once per workgroup through LDS).  Internal launchers are declared in launch.h.
#include "launch.h"
#include "enc_block16.h"
#include "enc_va16.h"

using namespace esmi;
ESMI_TU_RANGE_SETTER(enc_block16)
ESMI_TU_CHAIN_TRACE_SETTER(enc_block16)

namespace esmi {

// Whole encoder block in one launch, weight-folded attention (EncAttnFfnP::fold), one workgroup per utterance:
//   block 0 of tiny ES (C = 32, one head, k = 3 stride-1 merge conv folded into embedding tables), N <= 128;
//   block 1 of tiny ES (C = 64, two heads, k = 1 stride-2 merge conv from 32 channels), N <= 64.
// ESMI_ERR_UNSUPPORTED otherwise (-> launch_enc_block).  The split-f16 build only.
static bool b16_common_ok(const EncAttnFfnP& p) { return p.fold && p.N >= 1 && p.m.qkv_w && p.proj_w && p.ffn_w && p.mlp2_w; }
static bool b0_16_ok(const EncAttnFfnP& p) {
    return b16_common_ok(p) && p.C == 32 && p.h == 1 && p.m.k == 3 && p.m.stride == 1 && p.m.ids && p.m.emb_conv && p.N <= 128 && p.m.n_in == p.N;
}
static bool b1_16_ok(const EncAttnFfnP& p, int c_in) {
    return b16_common_ok(p) && p.C == 64 && p.h == 2 && p.m.k == 1 && p.m.stride == 2 && c_in == 32 && p.m.x_in && !p.m.ids && p.N <= 64 &&
           p.m.n_in >= 2 * p.N - 1;
}

#if ESMI_CHAIN_SPLIT
// The three chain16 kernels behind each other in one launch: producer and consumer of the block outputs are the same workgroup, so a
// drained store queue and a workgroup barrier are all that separates the stages (the rows travel through L2, as between launches).
struct EncAll16P { EncAttnFfnP b0, b1; FuseVaP va; };
template <int NKT>
__global__ __launch_bounds__(64 * 8, 2) void enc_all16_kernel(const EncAll16P p) {
    enc_b0_16_body<NKT>(p.b0);
    wait_vm0();
    __syncthreads();
    enc_b1_16_body(p.b1);
    wait_vm0();
    __syncthreads();
    enc_va16_body<3>(p.va);
}
#endif

int launch_enc_all16(const EncAttnFfnP& b0, const EncAttnFfnP& b1, int c_in1, const FuseVaP& va, int dim, int kernel, hipStream_t st) {
#if ESMI_CHAIN_SPLIT
    if (!b0_16_ok(b0) || !b1_16_ok(b1, c_in1) || !enc_va16_ok(va, dim, kernel)) return ESMI_ERR_UNSUPPORTED;
    if (va.T != b0.N || b0.B != b1.B || b0.B != va.B) return ESMI_ERR_UNSUPPORTED;
    // one wave count for the three bodies: the largest any of them needs, even (block 1 pairs its waves).  A body that needs fewer treats
    // the surplus waves' rows like the rows behind the end of a sequence (outside: zero inputs, nothing stored).
    int nw = (b0.N + 15) / 16;
    const int nw1 = 2 * ((b1.N + 15) / 16);
    nw = (nw > nw1 ? nw : nw1);
    nw += nw & 1;
    if (nw > 8) return ESMI_ERR_UNSUPPORTED;
    EncAll16P q;
    q.b0 = b0; q.b1 = b1; q.va = va;
    int lds = B016Lds::total > B116Lds::total ? B016Lds::total : B116Lds::total;
    lds = (lds > Va16Lds::total ? lds : Va16Lds::total) * (int)sizeof(float);
#define ESMI_ALL(NKT) { static AttrOnce once;                                                                              \
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(enc_all16_kernel<NKT>), once)) return rc;                \
        ESMI_LAUNCH((enc_all16_kernel<NKT>), dim3(b0.B), dim3(64 * nw), lds, st, q); return launch_status(); }
    if (nw <= 2) ESMI_ALL(2)
    if (nw <= 4) ESMI_ALL(4)
    ESMI_ALL(8)
#undef ESMI_ALL
#else
    (void)b0; (void)b1; (void)c_in1; (void)va; (void)dim; (void)kernel; (void)st;
    return ESMI_ERR_UNSUPPORTED;
#endif
}

int launch_enc_block16(const EncAttnFfnP& p, int expansion, int c_in, hipStream_t st) {
#if ESMI_CHAIN_SPLIT
    if (expansion != 1) return ESMI_ERR_UNSUPPORTED;
    EncAttnFfnP q = p;
    q.wgs_per_b = 1; q.halo = 0;
    if (b0_16_ok(p)) {
        const int nw = (p.N + 15) / 16, lds = B016Lds::total * (int)sizeof(float);
#define ESMI_B0(NKT) { static AttrOnce once;                                                                              \
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(enc_b0_16_kernel<NKT>), once)) return rc;               \
        ESMI_LAUNCH((enc_b0_16_kernel<NKT>), dim3(p.B), dim3(64 * nw), lds, st, q); return launch_status(); }
        if (nw <= 2) ESMI_B0(2)
        if (nw <= 4) ESMI_B0(4)
        ESMI_B0(8)
#undef ESMI_B0
    }
    if (b1_16_ok(p, c_in)) {
        const int nrt = (p.N + 15) / 16, lds = B116Lds::total * (int)sizeof(float);
        static AttrOnce once;
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(enc_b1_16_kernel), once)) return rc;
        ESMI_LAUNCH(enc_b1_16_kernel, dim3(p.B), dim3(128 * nrt), lds, st, q);
        return launch_status();
    }
    return ESMI_ERR_UNSUPPORTED;
#else
    (void)p; (void)expansion; (void)c_in; (void)st;
    return ESMI_ERR_UNSUPPORTED;
#endif
}

}  // namespace esmi
