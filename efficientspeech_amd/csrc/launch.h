// Host-side glue shared by the translation units of libesmi.so: stream / status helpers and the internal launchers
// (one per kernel family; each is defined in the one translation unit that instantiates that family's kernels, so the
// families compile in parallel and a kernel edit rebuilds one file).  Not part of the C-ABI (include/esmi.h is).
#ifndef ESMI_LAUNCH_H
#define ESMI_LAUNCH_H
#include "../../include/esmi.h"

#include <cmath>
#include <cstring>

#include "attention.h"
#include "convgemm.h"
#include "pwgemm.h"
#include "hifigan_resblock.h"
#include "enc_attn_ffn.h"
#include "enc_fuse_va.h"
#include "enc_merge_qkv.h"
#include "esmi_dev.h"
#include "small_kernels.h"

namespace esmi {

inline hipStream_t S(esmi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? ESMI_OK : (int)e;
}
inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: remember per (instantiation, device ordinal) that
// it was made (once: keeps the call out of hipGraph captures).  `done` is one static table per call site.
constexpr int kMaxDevices = 64;
struct AttrOnce { bool done[kMaxDevices] = {}; };
inline int raise_lds_limit(const void* fn, AttrOnce& once) {
    const int dev = current_device();
    if (dev < 0) return -dev;
    if (dev >= kMaxDevices) return ESMI_ERR_UNSUPPORTED;
    if (!once.done[dev]) {
        if (int rc = set_max_dynamic_lds(fn, 160 * 1024)) return rc;
        once.done[dev] = true;
    }
    return ESMI_OK;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int conv_out_len(int n, int k, int stride, int pad) { return (n + 2 * pad - k) / stride + 1; }
inline unsigned grid1d(long n, int block = 256) { return (unsigned)((n + block - 1) / block); }

// tu_convgemm.hip
ConvGemmP conv_defaults();
int launch_convgemm(ConvGemmP p, hipStream_t st);
// tu_attention.hip
int launch_attn(const AttnP& p, hipStream_t st);
// tu_enc_merge.hip / tu_enc_block.hip / tu_enc_attn_ffn.hip / tu_enc_fuse_va.hip (the encoder-side chain kernels)
int launch_enc_merge_qkv(const EncMergeP& p, int c_in, int c_out, hipStream_t st);
int launch_enc_block(const EncAttnFfnP& p, int expansion, int c_in, int plan, hipStream_t st);
int launch_enc_attn_ffn(const EncAttnFfnP& p, int expansion, int plan, hipStream_t st);
int launch_enc_fuse_va(const FuseVaP& p, int dim, int kernel, int nw, bool head, hipStream_t st);
// tu_enc_va16.hip (round 5: 16-row tiles, weights through LDS; ESMI_ERR_UNSUPPORTED for the shapes it is not built for)
int launch_enc_va16(const FuseVaP& p, int dim, int kernel, hipStream_t st);
bool enc_va16_ok(const FuseVaP& p, int dim, int kernel);
// tu_enc_va64.hip (round 6: the same stage for dim = 64 models, T <= 256; ESMI_ERR_UNSUPPORTED otherwise)
int launch_enc_va64(const FuseVaP& p, int dim, int kernel, hipStream_t st);
struct PostAttn64P;
int launch_enc_post_attn64(const PostAttn64P& p, hipStream_t st);   // (enc_ffn64.h: proj + LN1 + MixFFN + LN2 of a C = 64 one-head block, N <= 256)
// tu_enc_pred128.hip (round 6: the three predictors + variance-adaptor tail + scan of a dim = 128 model, T <= 256; reads feat[:, 0 .. dim))
int launch_enc_pred128(const Pred128P& p, int dim, hipStream_t st);
int launch_enc_fuse128(const FuseVaP& p, int dim, int kernel, hipStream_t st);
int launch_enc_post_attn128(const PostAttn128P& p, hipStream_t st);
int launch_enc_merge_q256(const MergeQ256P& p, hipStream_t st);   // (enc_merge256.h: merge conv k = 5 stride 2, 128 -> 256, + the folded query GEMM, N <= 128)   // (enc_ffn128.h: proj + LN1 + MixFFN + LN2 of a C = 128 two-head expansion-2 block, N <= 256)
   // (enc_fuse128.h: the Fuse stage of the same models, one launch)
// tu_enc_block16.hip (round 5: whole-block kernels of dim = 32 models on 16-row tiles)
int launch_enc_block16(const EncAttnFfnP& p, int expansion, int c_in, hipStream_t st);
// ... and the whole encoder side in one launch (block 0 | block 1 | Fuse + variance adaptor), when each of the three chain16 kernels
// serves its shape and they run the same number of waves; ESMI_ERR_UNSUPPORTED otherwise
int launch_enc_all16(const EncAttnFfnP& b0, const EncAttnFfnP& b1, int c_in1, const FuseVaP& va, int dim, int kernel, hipStream_t st);
// tu_hifigan.hip
int launch_resblock(const ResblockP& p, int c, hipStream_t st);

inline bool enc_attn_ffn_supported(int C, int N, int expansion) {
    if ((C & 31) || N > 256 || N < 1) return false;
    // sequences of more than 128 positions: the chain kernel runs one latency chain per 32 rows against up to 256 keys; the same
    // ops as launches (LDS-staged attention + LDS-staged GEMMs) are faster there (small ES T = 256: 2.53 vs 2.62 ms/step; base ES
    // block 0: 1.33 vs 2.00 ms) -- `tools/debug_plan_base.py` measures the plans
    if (N > 128) return false;
    const int nc = C / 32;
    // base ES block 0 (C = 128, expansion 2) at N = 256: the chain kernel runs one latency chain per 32 rows against 256 keys
    // (2.00 ms at B = 512); the same ops as LDS-staged GEMM launches take 1.33 ms, so that shape goes per-op
    return (expansion == 1 && (nc == 1 || nc == 2 || nc == 4)) || (expansion == 2 && nc == 4 && N <= 128);
}



// ---- activation-range check (esmi_dev.h, the ESMI_RANGE_CHECK build): every translation unit owns a copy of the device-side flag
// pointer and defines its setter with ESMI_TU_RANGE_SETTER(<unit>); `set_range_flag_all` (esmi_abi.hip) calls them all.
int set_range_flag_abi(int* flag);
int set_range_flag_convgemm(int* flag);
int set_range_flag_attention(int* flag);
int set_range_flag_enc_merge(int* flag);
int set_range_flag_enc_block(int* flag);
int set_range_flag_enc_attn_ffn(int* flag);
int set_range_flag_enc_fuse_va(int* flag);
int set_range_flag_enc_va16(int* flag);
int set_range_flag_enc_va64(int* flag);
int set_range_flag_enc_pred128(int* flag);
int set_range_flag_enc_block16(int* flag);
int set_range_flag_decoder(int* flag);
int set_range_flag_dec_128_5(int* flag);
int set_range_flag_dec_128_3(int* flag);
int set_range_flag_dec_256_5(int* flag);
int set_range_flag_dec_256_3(int* flag);
int set_range_flag_hifigan(int* flag);
int set_range_flag_train(int* flag);
// development (-DESMI_CHAIN_TRACE): every translation unit with chain kernels owns a copy of the trace pointer
#ifdef ESMI_CHAIN_TRACE
#define ESMI_TU_CHAIN_TRACE_SETTER(tu) extern "C" void esmi_dev_set_chain_trace_##tu(long long* ptr) { \
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chain_trace_dev), &ptr, sizeof(ptr)); }
#else
#define ESMI_TU_CHAIN_TRACE_SETTER(tu)
#endif
#if ESMI_RANGE_CHECK
#define ESMI_TU_RANGE_SETTER(tu) namespace esmi { int set_range_flag_##tu(int* flag) { return store_range_flag_pointer(flag); } }
#else
#define ESMI_TU_RANGE_SETTER(tu) namespace esmi { int set_range_flag_##tu(int*) { return ESMI_OK; } }
#endif

}  // namespace esmi
#endif  // ESMI_LAUNCH_H
