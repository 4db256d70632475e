// esmi C-ABI, translation unit "tu_dec_128_5.hip": mel_decoder_kernel<128, 5, NW> (mel_decoder.h) and its launcher -- one
// instantiation per file: this kernel dominates the library's compile time, so the four build side by side.
#include "launch.h"
#include "mel_decoder.h"

#ifndef ESMI_DEC_NW256
#define ESMI_DEC_NW256 8    // waves per window of the dx2 = 256 decoder (small / base ES): 8, or 16 (measured 30 % slower: 65 spilled
#endif                      // VGPRs at the 128-register budget and twice the weight traffic; small ES decoder 2.43 vs 1.87 ms)

ESMI_TU_RANGE_SETTER(dec_128_5)

namespace esmi {

int launch_mel_decoder_128_5(const MelDecP& p, dim3 grid, hipStream_t st) {
    constexpr int DX2 = 128, KD = 5, NW = DX2 == 128 ? 8 : ESMI_DEC_NW256;
    const int lds = dec_lds_floats<DX2>(KD) * (int)sizeof(float);
    static AttrOnce once;
    if (int rc = raise_lds_limit(reinterpret_cast<const void*>(mel_decoder_kernel<DX2, KD, NW>), once)) return rc;
    ESMI_LAUNCH((mel_decoder_kernel<DX2, KD, NW>), grid, dim3(64 * NW), lds, st, p);
    return launch_status();
}

}  // namespace esmi
