// esmi C-ABI, translation unit "tu_dec_pp.hip": mel_decoder_pp_kernel<KD> (mel_decoder_pp.h) and its launcher -- the dx2 = 128
// decoder with a supplied first stage (what the tiny model's one-call forward runs).
#include "launch.h"
#include "mel_decoder_pp.h"

ESMI_TU_RANGE_SETTER(dec_pp)

namespace esmi {

// One persistent workgroup per CU (MI355X: 256 CUs in 8 XCDs); fewer when there are fewer than two windows per workgroup.
int launch_mel_decoder_pp(const MelDecP& p, int kernel, hipStream_t st) {
    const int nq_max = ((p.B + 7) / 8) * p.n_tiles;            // windows of the busiest XCD
    const int wg_per_xcd = std::min(32, std::max(1, (nq_max + 1) / 2));
    const dim3 grid((unsigned)(8 * wg_per_xcd)), block(kPpThreads);
    static AttrOnce once5, once3;
    if (kernel == 5) {
        const int lds = dec_pp_lds_floats<5>() * (int)sizeof(float);
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(mel_decoder_pp_kernel<5>), once5)) return rc;
        ESMI_LAUNCH((mel_decoder_pp_kernel<5>), grid, block, lds, st, p);
    } else {
        const int lds = dec_pp_lds_floats<3>() * (int)sizeof(float);
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(mel_decoder_pp_kernel<3>), once3)) return rc;
        ESMI_LAUNCH((mel_decoder_pp_kernel<3>), grid, block, lds, st, p);
    }
    return launch_status();
}

}  // namespace esmi
