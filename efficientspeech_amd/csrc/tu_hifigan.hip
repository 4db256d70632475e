// esmi C-ABI, translation unit "tu_hifigan.hip": HiFi-GAN ResBlock kernels and their weight packer (hifigan_resblock.h)
// One of several translation units of libesmi.so (compiled in parallel by __graft_entry__.build(); the simulator build
// tools/wavesim/build.sh compiles the same files with the host compiler).  Internal launchers are declared in launch.h.
#include "launch.h"

using namespace esmi;
ESMI_TU_RANGE_SETTER(hifigan)

namespace esmi {

// One ResBlock in one launch (hifigan_resblock.h) when its packed weights are there and (channels, kernel size) has an
// instantiation; `false` from resblock_fused_ok -> the caller runs the block conv by conv.
template <int C, int K>
int launch_resblock_ck(const ResblockP& p, hipStream_t st) {
    const size_t lds = rb_lds_bytes(C, p.R);
    const dim3 grid((unsigned)(p.B * p.tiles_per_b)), block(64 * kRbWaves);
    if constexpr (C <= 16) {   // narrow MFMA tiles (16 channels x 16 positions): LDS <= 32 KB, no limit to raise
        ESMI_LAUNCH((hifigan_resblock16_kernel<C, K>), grid, block, lds, st, p);
    } else {
        static AttrOnce once;
        if (lds > 48 * 1024)
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(hifigan_resblock_kernel<C, K>), once)) return rc;
        ESMI_LAUNCH((hifigan_resblock_kernel<C, K>), grid, block, lds, st, p);
    }
    return launch_status();
}
template <int C>
int launch_resblock_c(const ResblockP& p, hipStream_t st) {
    switch (p.k) {
        case 3: return launch_resblock_ck<C, 3>(p, st);
        case 7: return launch_resblock_ck<C, 7>(p, st);
        case 11: return launch_resblock_ck<C, 11>(p, st);
    }
    return ESMI_ERR_UNSUPPORTED;
}
int launch_resblock(const ResblockP& p, int c, hipStream_t st) {
    switch (c) {
        case 8: return launch_resblock_c<8>(p, st);
        case 16: return launch_resblock_c<16>(p, st);
        case 32: return launch_resblock_c<32>(p, st);
        case 64: return launch_resblock_c<64>(p, st);
    }
    return ESMI_ERR_UNSUPPORTED;
}

}  // namespace esmi

extern "C" {

size_t esmi_pack_resblock_bytes(int c, int k) {
    if ((c != 8 && c != 16 && c != 32 && c != 64) || (k != 3 && k != 7 && k != 11)) return 0;
    return rb_pack_dwords(c, k) * 4;
}
int esmi_pack_resblock_f16(const float* src, void* dst, int c, int k, esmi_stream_t stream) {
    if (!src || !dst) return ESMI_ERR_ARG;
    if (!esmi_pack_resblock_bytes(c, k)) return ESMI_ERR_UNSUPPORTED;
    if (c <= 16) {
        const long n16 = (long)rb_ksteps16(c, k) * 64;
        ESMI_LAUNCH(pack_resblock16_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, S(stream), src, static_cast<unsigned*>(dst), c, k);
        return launch_status();
    }
    const long n = (long)rb_mtiles(c) * rb_ksteps(c, k) * 64;
    ESMI_LAUNCH(pack_resblock_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), src, static_cast<unsigned*>(dst), c, k);
    return launch_status();
}

}  // extern "C"
