// esmi C-ABI, translation unit "tu_convgemm.hip": implicit-GEMM convolutions / Linears of the one-kernel-per-op plan (convgemm.h)
// One of several translation units of libesmi.so (compiled in parallel by __graft_entry__.build(); the simulator build
// tools/wavesim/build.sh compiles the same files with the host compiler).  Internal launchers are declared in launch.h.
#include "launch.h"

using namespace esmi;
ESMI_TU_RANGE_SETTER(convgemm)


// rows (B * n_out) from which the per-op plan's GEMMs take the LDS-staged kernel: kGemmLdsMinRowsDefault (wavesim_shim.h: 2048 on the
// GPU, 1 in the simulator)

#ifdef ESMI_GEMM_TRACE
namespace esmi { __device__ long long* g_gemm_trace_dev = nullptr; }
extern "C" void esmi_dev_set_gemm_trace(long long* ptr) { hipMemcpyToSymbol(HIP_SYMBOL(esmi::g_gemm_trace_dev), &ptr, sizeof(ptr)); }
#endif

namespace esmi {

ConvGemmP conv_defaults() {
    ConvGemmP p;
    memset(&p, 0, sizeof p);
    p.k = 1; p.stride = 1; p.pad = 0; p.mode = MODE_CONV;
    return p;
}

// full_row: the epilogue needs a whole output row inside one wave (LayerNorm / row-dot)
int launch_convgemm(ConvGemmP p, hipStream_t st) {
    if ((p.c_in & 7) || p.c_in <= 0 || p.c_out <= 0 || p.n_out <= 0 || p.B <= 0) return ESMI_ERR_ARG;
    // (weights: the fp32 tensor, or -- split-f16 build, c_in a multiple of 32 -- the pre-split blob alone: every kernel below that is
    // reached then reads the blob only)
    const bool blob_only = !p.W && ESMI_CHAIN_SPLIT && p.Wp && aligned16(p.Wp) && (p.c_in & 31) == 0 && p.c_out > 1;
    if (!blob_only && (!p.W || !aligned16(p.W))) return ESMI_ERR_ARG;
    if (p.ids) {
        if (!p.table || (p.ld_table & 3) || !aligned16(p.table)) return ESMI_ERR_ARG;
    } else if (!p.A || (p.lda & 3) || (p.a_coff & 3) || !aligned16(p.A)) return ESMI_ERR_ARG;
    const bool full_row = p.ln_g || p.dot_out;
    if (p.c_out == 1 && p.mode == MODE_CONV && p.stride == 1 && !p.ids && !full_row && !p.res && !p.rowmask && p.out) {
        const long n = (long)p.B * p.n_out;
        ESMI_LAUNCH(conv_to1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
        return launch_status();
    }
    int nt;
    if (full_row) {
        nt = (p.c_out + 31) / 32;
        if (nt == 3) nt = 4;
        if (nt > 4 && nt <= 8) nt = 8;
        if (nt > 8) return ESMI_ERR_UNSUPPORTED;
        if (p.ln_g && p.c_out != 32 * nt) return ESMI_ERR_UNSUPPORTED;  // LN width must be 32/64/128/256
    } else {
        nt = p.c_out > 64 ? 4 : (p.c_out > 32 ? 2 : 1);
        // small problems: narrower column tiles until the launch has ~a wave per SIMD.  A wave splits its OWN 32 nt weight rows into
        // the f16 pieces at every k-step (this kernel takes the weights as stored), so with 200-400 row tiles -- the training step's
        // encoder-size GEMMs, 64 workgroups on 256 CUs -- a wave was VALU-bound on splitting 2-4 column tiles' weights while three
        // quarters of the chip idled.  An output element's products and their order do not depend on nt: bitwise the same results.
        const long row_tiles = (long)p.B * convgemm_tiles_per_phase(p) * convgemm_row_stride(p);
        while (nt > 1 && row_tiles * ((p.c_out + 32 * nt - 1) / (32 * nt)) < 1024) nt >>= 1;
    }
#if ESMI_CHAIN_SPLIT
    // a Linear over many rows whose whole weight fits in LDS (the training step's decoder GEMMs): pwgemm.h -- one workgroup per CU
    // stages the weight once and its waves walk the rows
    if (p.pw_ok && p.mode == MODE_CONV && p.k == 1 && p.stride == 1 && !p.ids && !p.act_in && (p.c_in == 128 || p.c_in == 80) && p.c_out > 64 && p.c_out <= 128 &&
        (!p.ln_g || p.c_out == 128) && p.n_in == p.n_out && (long)p.B * p.n_out >= kPwGemmMinRows &&
        (((long)p.B * p.n_in) * p.lda + p.a_coff + p.c_in) * 4L < (1L << 31)) {
        const long rows = (long)p.B * p.n_out;
        // a wave's item: 32 rows x 64 columns (two per row tile: reads and stores of neighbouring items overlap), or x all columns
        // when the epilogue needs whole rows in one wave
        const int nh = full_row ? 1 : 2, n_items = (int)((rows + 31) / 32) * nh, want = (n_items + kPwWaves - 1) / kPwWaves;
        const dim3 grid((unsigned)(want < 256 ? want : 256));
        static AttrOnce once[8];
#define ESMI_PW_CASE(KS_, NT_, AMP_, slot)                                                                                     \
    do {                                                                                                                       \
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(pwgemm_kernel<KS_, NT_, AMP_>), once[slot])) return rc;        \
        ESMI_LAUNCH((pwgemm_kernel<KS_, NT_, AMP_>), grid, dim3(64 * kPwWaves), pwgemm_lds_bytes<KS_>(), st, p, n_items);        \
    } while (0)
        const int variant = (p.c_in == 128 ? 0 : 4) + (full_row ? 2 : 0) + (p.amp ? 1 : 0);
        switch (variant) {
            case 0: ESMI_PW_CASE(8, 2, false, 0); break;
            case 1: ESMI_PW_CASE(8, 2, true, 1); break;
            case 2: ESMI_PW_CASE(8, 4, false, 2); break;
            case 3: ESMI_PW_CASE(8, 4, true, 3); break;
            case 4: ESMI_PW_CASE(5, 2, false, 4); break;
            case 5: ESMI_PW_CASE(5, 2, true, 5); break;
            case 6: ESMI_PW_CASE(5, 4, false, 6); break;
            default: ESMI_PW_CASE(5, 4, true, 7); break;
        }
#undef ESMI_PW_CASE
        return launch_status();
    }
    // large plain convolutions / Linears: operands staged through LDS by convgemm_dma_kernel (convgemm.h) -- needs input rows ==
    // output rows (flat-row addressing) and 32-bit lane offsets: the kernel forms (row + tile rows + halo) * lda BEFORE it clamps, so
    // the bound covers the last workgroup's padded rows and the taps' reach, not just the tensor (the pre-clamp product
    // must not overflow); anything else streams from L2 below
    if (p.mode == MODE_CONV && p.stride == 1 && !p.ids && (p.c_in & 31) == 0 && p.c_out > 64 && (long)p.B * p.n_out >= kGemmLdsMinRowsDefault &&
        p.n_in == p.n_out &&
        (((long)p.B * p.n_in + 64L * kGemmLdsWaves + 2L * kGemmHaloMax + 8) * p.lda + p.a_coff + p.c_in) < (1L << 31)) {
        constexpr int NWV = kGemmLdsWaves;
        const long rows = (long)p.B * p.n_out;
        // a LayerNorm / row-dot epilogue over 129..256 channels needs them all in one wave: 32 rows x 256 channels per wave; else
        // 128 channels per wave and 64 rows when that still gives every CU its two workgroups, 32 otherwise (training at phoneme rate)
        const bool wide = full_row && nt > 4;
        const int ny = full_row ? 1 : (p.c_out + 127) / 128;
        const int mt = (!wide && ((rows + 64 * NWV - 1) / (64 * NWV)) * ny >= 512) ? 2 : 1;
        const long wr = 32 * mt * NWV;
        const int nx = (int)((rows + wr - 1) / wr);
        dim3 g1((unsigned)((nx + 7) / 8 * 8 * ny));
        const bool pre = p.Wp != nullptr && aligned16(p.Wp);
        const int lds_bytes = wide ? convgemm_dma_bytes<8>(1, p.k, p.dil, pre) : convgemm_dma_bytes<4>(mt, p.k, p.dil, pre);
        static AttrOnce once[12];
#define ESMI_DMA_CASE(NT_, MT_, AMP_, PRE_, slot)                                                                              \
    do {                                                                                                                       \
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(convgemm_dma_kernel<NT_, MT_, NWV, AMP_, PRE_>), once[slot])) return rc; \
        ESMI_LAUNCH((convgemm_dma_kernel<NT_, MT_, NWV, AMP_, PRE_>), g1, dim3(64 * NWV), lds_bytes, st, p, nx, ny);           \
    } while (0)
        const int variant = (wide ? 8 : (mt == 2 ? 4 : 0)) + (p.amp ? 2 : 0) + (pre ? 1 : 0);
        switch (variant) {
            case 0: ESMI_DMA_CASE(4, 1, false, false, 0); break;
            case 1: ESMI_DMA_CASE(4, 1, false, true, 1); break;
            case 2: ESMI_DMA_CASE(4, 1, true, false, 2); break;
            case 3: ESMI_DMA_CASE(4, 1, true, true, 3); break;
            case 4: ESMI_DMA_CASE(4, 2, false, false, 4); break;
            case 5: ESMI_DMA_CASE(4, 2, false, true, 5); break;
            case 6: ESMI_DMA_CASE(4, 2, true, false, 6); break;
            case 7: ESMI_DMA_CASE(4, 2, true, true, 7); break;
            case 8: ESMI_DMA_CASE(8, 1, false, false, 8); break;
            case 9: ESMI_DMA_CASE(8, 1, false, true, 9); break;
            case 10: ESMI_DMA_CASE(8, 1, true, false, 10); break;
            default: ESMI_DMA_CASE(8, 1, true, true, 11); break;
        }
#undef ESMI_DMA_CASE
        return launch_status();
    }
#endif
    const int tiles = p.B * convgemm_tiles_per_phase(p) * convgemm_row_stride(p);
    dim3 grid((tiles + 3) / 4, full_row ? 1 : (p.c_out + 32 * nt - 1) / (32 * nt));
    dim3 block(256);
    if (!p.amp) {
        switch (nt) {
            case 1: ESMI_LAUNCH((convgemm_kernel<1, false>), grid, block, 0, st, p); break;
            case 2: ESMI_LAUNCH((convgemm_kernel<2, false>), grid, block, 0, st, p); break;
            case 4: ESMI_LAUNCH((convgemm_kernel<4, false>), grid, block, 0, st, p); break;
            case 8: ESMI_LAUNCH((convgemm_kernel<8, false>), grid, block, 0, st, p); break;
            default: return ESMI_ERR_UNSUPPORTED;
        }
    } else {
        switch (nt) {
            case 1: ESMI_LAUNCH((convgemm_kernel<1, true>), grid, block, 0, st, p); break;
            case 2: ESMI_LAUNCH((convgemm_kernel<2, true>), grid, block, 0, st, p); break;
            case 4: ESMI_LAUNCH((convgemm_kernel<4, true>), grid, block, 0, st, p); break;
            case 8: ESMI_LAUNCH((convgemm_kernel<8, true>), grid, block, 0, st, p); break;
            default: return ESMI_ERR_UNSUPPORTED;
        }
    }
    return launch_status();
}

}  // namespace esmi
