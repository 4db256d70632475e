// esmi C-ABI, translation unit "tu_dec_256_5.hip": mel_decoder_kernel<256, 5, NW> (mel_decoder.h) and its launcher -- one
// instantiation per file: this kernel dominates the library's compile time, so the four build side by side.
#include "launch.h"
#include "mel_decoder.h"

ESMI_TU_RANGE_SETTER(dec_256_5)

namespace esmi {

int set_dec_clock_256_5(long long* slots) { return store_dec_clock_pointer(slots); }

int launch_mel_decoder_256_5(const MelDecP& p, dim3 grid, hipStream_t st) {
    constexpr int DX2 = 256, KD = 5, NW = 8;   // waves per window (16 for dx2 = 256 measured 30 % slower: HISTORY.md 3.1)
    const int lds = (dec_lds_floats<DX2>(KD) + p.carry_lds_layers * (KD / 2) * DX2) * (int)sizeof(float);
    static AttrOnce once;
    if (int rc = raise_lds_limit(reinterpret_cast<const void*>(mel_decoder_kernel<DX2, KD, NW>), once)) return rc;
    ESMI_LAUNCH((mel_decoder_kernel<DX2, KD, NW>), grid, dim3(64 * NW), lds, st, p);
    return launch_status();
}

}  // namespace esmi
