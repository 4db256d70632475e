// esmi C-ABI, translation unit "tu_decoder.hip": the mel decoder: blob packer and mel_decoder_kernel launches (mel_decoder.h)
// One of several translation units of libesmi.so (compiled in parallel by __graft_entry__.build(); the simulator build
// tools/wavesim/build.sh compiles the same files with the host compiler).  Internal launchers are declared in launch.h.
#include "launch.h"
#include <cstdlib>

#include "mel_decoder.h"

using namespace esmi;
ESMI_TU_RANGE_SETTER(decoder)

namespace esmi {
int launch_mel_decoder_128_5(const MelDecP& p, dim3 grid, hipStream_t st);
int launch_mel_decoder_128_3(const MelDecP& p, dim3 grid, hipStream_t st);
int launch_mel_decoder_256_5(const MelDecP& p, dim3 grid, hipStream_t st);
int launch_mel_decoder_256_3(const MelDecP& p, dim3 grid, hipStream_t st);
int set_dec_clock_128_5(long long* slots);
int set_dec_clock_128_3(long long* slots);
int set_dec_clock_256_5(long long* slots);
int set_dec_clock_256_3(long long* slots);
}  // namespace esmi

#ifdef ESMI_DEC_TRACE
long long* g_esmi_trace = nullptr;
extern "C" void esmi_dev_set_trace(long long* ptr) { g_esmi_trace = ptr; }
#endif

extern "C" {

// measurement aid (include/esmi.h): arm / disarm the clock probe of every decoder instantiation (one device global per unit)
int esmi_mel_decoder_clock_probe(int64_t* dev_slots) {
    long long* s = reinterpret_cast<long long*>(dev_slots);
    int (*const setters[])(long long*) = {set_dec_clock_128_5, set_dec_clock_128_3, set_dec_clock_256_5, set_dec_clock_256_3};
    for (auto set : setters)
        if (int rc = set(s)) return rc;
    return ESMI_OK;
}

static int dec_check(const esmi_decoder_shape* s) {
    if (!s) return ESMI_ERR_ARG;
    if (s->dx2 != 128 && s->dx2 != 256) return ESMI_ERR_UNSUPPORTED;
    if (s->d4 <= 0 || s->d4 % 128) return ESMI_ERR_UNSUPPORTED;
    if (s->kernel != 3 && s->kernel != 5) return ESMI_ERR_UNSUPPORTED;
    if (s->n_mel <= 0 || s->n_mel > kMelCols) return ESMI_ERR_UNSUPPORTED;
    if (s->n_blocks < 1 || s->block_depth < 1 || s->n_blocks * s->block_depth > ESMI_MAX_DEC_LAYERS) return ESMI_ERR_UNSUPPORTED;
    if (2 * (s->kernel / 2) * s->n_blocks * s->block_depth >= kDecRows - 32) return ESMI_ERR_UNSUPPORTED;
    return ESMI_OK;
}

size_t esmi_mel_decoder_blob_bytes(const esmi_decoder_shape* s) {
    if (dec_check(s)) return 0;
    return (size_t)dec_layout(s->d4, s->dx2, s->kernel, s->n_blocks, s->block_depth).total * sizeof(float);
}

int esmi_mel_decoder_pack_f32(const esmi_decoder_weights* w, const esmi_decoder_shape* s, float* blob,
                              esmi_stream_t stream) {
    int rc = dec_check(s);
    if (rc) return rc;
    if (!w || !blob) return ESMI_ERR_ARG;
    const DecLayout L = dec_layout(s->d4, s->dx2, s->kernel, s->n_blocks, s->block_depth);
    hipStream_t st = S(stream);
    const int dx2 = s->dx2, ntw = dx2 / 128;
    auto bslice = [&](const float* src, long off, int N, int K, int ntw) {
#if ESMI_DEC_SPLIT == 2
        const long n = (long)(K / 128) * 4 * ntw * 8 * 2 * 256;
        ESMI_LAUNCH(pack_bslice2h_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src,
                    reinterpret_cast<unsigned*>(blob + off), N, K, ntw);
#else
        const long n = (long)(K / 128) * 4 * ntw * 16 * 256;
        ESMI_LAUNCH(pack_bslice_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, blob + off, N, K, ntw);
#endif
    };
    auto vec = [&](const float* src, long off, int n, int n_pad) {
        ESMI_LAUNCH(copy_pad_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, st, src, blob + off, n, n_pad);
    };
    bslice(w->proj_w, L.proj_w, dx2, s->d4, ntw);
    vec(w->proj_b, L.proj_b, dx2, dx2);
    vec(w->proj_ln_g, L.proj_g, dx2, dx2);
    vec(w->proj_ln_b, L.proj_beta, dx2, dx2);
    for (int l = 0; l < s->n_blocks * s->block_depth; ++l) {
        const long base = L.layer0 + (long)l * L.layer_stride;
        if (!w->dw_w[l] || !w->pw_w[l]) return ESMI_ERR_ARG;
        ESMI_LAUNCH(pack_dw_kernel, dim3((dx2 * s->kernel + 255) / 256), dim3(256), 0, st, w->dw_w[l], blob + base + L.l_dw,
                    dx2, s->kernel);
        vec(w->dw_b[l], base + L.l_dwb, dx2, dx2);
        bslice(w->pw_w[l], base + L.l_pw, dx2, dx2, ntw);
        vec(w->pw_b[l], base + L.l_pwb, dx2, dx2);
        vec(w->ln_g[l], base + L.l_g, dx2, dx2);
        vec(w->ln_b[l], base + L.l_b, dx2, dx2);
    }
    for (int b = 0; b < s->n_blocks; ++b) {
        vec(w->skip_g[b], L.skip0 + 2L * dx2 * b, dx2, dx2);
        vec(w->skip_b[b], L.skip0 + 2L * dx2 * b + dx2, dx2, dx2);
    }
    // (the mel Linear of the split build's dx2 = 256 kernel: three one-tile column slices, mel_decoder.h NTM)
#if ESMI_DEC_SPLIT == 2
    bslice(w->mel_w, L.mel_w, s->n_mel, dx2, 1);
#else
    bslice(w->mel_w, L.mel_w, s->n_mel, dx2, ntw);
#endif
    vec(w->mel_b, L.mel_b, s->n_mel, dx2);
    return launch_status();
}

// segments per utterance / frames per segment of the dx2 = 256 kernel when it may carry rows between chunks: whole utterances when
// the batch fills the chip, else as many segments per utterance as it takes to give every CU one (a segment's first chunk recomputes
// its left halo, so fewer, longer segments are cheaper)
// Block skew (mel_decoder.h, the chunk loop): a chunk advances by 128 - block_depth * k/2 frames instead of 128 - halo; used when the
// model has more than one block and a block carry fits one float4 per thread.
static bool stream_skew(const esmi_decoder_shape* s) {
    return s->n_blocks >= 2 && (s->block_depth + 1) * (s->kernel / 2) * (s->dx2 / 4) <= kDecBlockCarry4;
}
static void stream_geometry(const esmi_decoder_shape* s, int B, int L_out, int* n_seg_out, int* seg_len_out) {
    const int pad = s->kernel / 2, halo = pad * s->n_blocks * s->block_depth;
    const bool skew = stream_skew(s);
    const int sh = pad * s->block_depth;
    // frames a chunk keeps; frames in front of an utterance's first output frame (the last block's tile starts that much early); rows a
    // segment that starts inside an utterance recomputes (nothing is carried into its first chunk)
    const int keep = skew ? kDecRows - sh : kDecRows - halo, lead = skew ? halo - sh : 0, lost = skew ? 2 * halo - sh : halo;
    const int chunks = (L_out + lead + keep - 1) / keep;
    // (ESMI_DEC_STREAM_WGS: test knob -- the workgroup count the segmentation aims at, default one per CU; 1 = whole-utterance walks
    // at any batch size, which is how the CPU simulator tests reach multi-chunk segments without a 256-utterance batch)
    const char* env = getenv("ESMI_DEC_STREAM_WGS");
    const int target = env && atoi(env) > 0 ? atoi(env) : 256;
    int n_seg = (target + B - 1) / B;
    n_seg = n_seg < 1 ? 1 : (n_seg > chunks ? chunks : n_seg);
    int per = (chunks + n_seg - 1) / n_seg;                 // chunks per segment
    while (n_seg > 1 && per * keep - lost < keep) ++per;    // (short utterances: a later segment must still yield a chunk's worth)
    *seg_len_out = n_seg == 1 ? L_out : per * keep - lost;
    *n_seg_out = (L_out + *seg_len_out - 1) / *seg_len_out;
}
// scratch floats per workgroup: one k/2-row set per conv layer, then one block carry per block boundary
static int stream_ws_stride(const esmi_decoder_shape* s) {
    return s->n_blocks * s->block_depth * (s->kernel / 2) * s->dx2 + (s->n_blocks - 1) * 4 * kDecBlockCarry4;
}

size_t esmi_mel_decoder_workspace_bytes(const esmi_decoder_shape* s, int B, int L_out) {
    if (dec_check(s) || s->dx2 != 256 || B <= 0 || L_out == 0) return 0;
    int n_seg, seg_len;
    if (L_out < 0) {
        // any output length (the caller sizes its scratch before the length is known): the segmentation aims at max(B, one workgroup
        // per CU) segments and its rounding (whole chunks per segment, the rows a later segment recomputes) stays below 1.15x that
        stream_geometry(s, B, 1 << 20, &n_seg, &seg_len);
        n_seg = n_seg + n_seg / 4 + 1;
    } else {
        stream_geometry(s, B, L_out, &n_seg, &seg_len);
    }
    return (size_t)n_seg * B * stream_ws_stride(s) * sizeof(float);   // one row set per (segment, utterance)
}

static int mel_decoder_launch(const float* blob, const esmi_decoder_shape* s, const float* x, const float* h0,
                              const int32_t* cum, const int32_t* mel_len, const int32_t* lmax_dev, int lmax_host, int apply_mask, int B, int T,
                              int L_out, float* mel, void* workspace, size_t workspace_bytes, esmi_stream_t stream) {
    int rc = dec_check(s);
    if (rc) return rc;
    if (!blob || (!x && !h0) || !mel || B <= 0 || L_out <= 0 || !aligned16(blob) || (x && !aligned16(x))) return ESMI_ERR_ARG;
    if (h0 && (!cum || !aligned16(h0))) return ESMI_ERR_ARG;   // the phoneme-rate head only exists in the fused-gather mode
    if (!cum && lmax_dev) return ESMI_ERR_ARG;  // direct mode: L is the tensor's own length, known to the host
    if (!lmax_dev && lmax_host == 0) return ESMI_ERR_ARG;
    if (!lmax_dev && lmax_host < 0 && (!mel_len || !cum)) return ESMI_ERR_ARG;   // L derived from mel_len
    MelDecP p;
    p.blob = blob;
    {   // the kernel derives the blob layout itself (DecLay: compile-time offsets); it must be the one the packer wrote by
        const DecLayout L = dec_layout(s->d4, s->dx2, s->kernel, s->n_blocks, s->block_depth);
        const bool same = s->dx2 == 128 ? (s->kernel == 5 ? DecLay<128, 5>(s->d4, s->n_blocks, s->block_depth).matches(L) : DecLay<128, 3>(s->d4, s->n_blocks, s->block_depth).matches(L))
                                        : (s->kernel == 5 ? DecLay<256, 5>(s->d4, s->n_blocks, s->block_depth).matches(L) : DecLay<256, 3>(s->d4, s->n_blocks, s->block_depth).matches(L));
        if (!same || L.total > 0x1fffffffL) return ESMI_ERR_UNSUPPORTED;
    }
    p.d4 = s->d4; p.n_blocks = s->n_blocks; p.block_depth = s->block_depth; p.n_mel = s->n_mel;
    p.x = x; p.h0 = h0; p.cum = cum; p.mel_len = mel_len; p.lmax_dev = lmax_dev; p.lmax_host = lmax_host;
    p.apply_mask = apply_mask && mel_len; p.B = B; p.T = T; p.L_out = L_out; p.mel = mel;
    p.halo = (s->kernel / 2) * s->n_blocks * s->block_depth;
    p.trace = nullptr;
#ifdef ESMI_DEC_TRACE
    p.trace = g_esmi_trace;   // development only, see tools/dec_budget.py
#endif
    hipStream_t st = S(stream);
    p.carry_ws = nullptr;
    p.carry_lds_layers = 0;
    p.skew = 0;
    p.ws_stride = 0;
    const size_t need = esmi_mel_decoder_workspace_bytes(s, B, L_out);
    if (s->dx2 == 256 && workspace && need && workspace_bytes >= need) {
        // one workgroup per CU walks a segment chunk by chunk, each conv layer's rows in front of a chunk carried in `workspace`
        stream_geometry(s, B, L_out, &p.n_seg, &p.seg_len);
        p.carry_ws = static_cast<float*>(workspace);
        p.ws_stride = stream_ws_stride(s);
        p.skew = stream_skew(s) ? 1 : 0;
        // conv-layer carry slots: with the skew the first layer of every later block takes its pad rows from the block carry
        p.carry_lds_layers = dec_carry_lds_layers<256>(s->kernel, s->n_blocks * s->block_depth - (p.skew ? s->n_blocks - 1 : 0));
    } else {
        // every 128-row window is its own segment, halo rows recomputed on both sides (dx2 = 128: two workgroups per CU balance
        // the chip and 768 = 7 x 112 frames leaves nothing to gain; dx2 = 256 without a workspace)
        p.seg_len = kDecRows - 2 * p.halo;
        p.n_seg = (L_out + p.seg_len - 1) / p.seg_len;
    }
    dim3 grid((unsigned)(p.n_seg * ((B + 7) / 8) * 8)), block(kDecThreads);
    // one translation unit per instantiation (tu_dec_<dx2>_<k>.hip): the kernel is by far the slowest thing to compile
    if (s->dx2 == 128 && s->kernel == 5) return launch_mel_decoder_128_5(p, grid, st);
    if (s->dx2 == 128 && s->kernel == 3) return launch_mel_decoder_128_3(p, grid, st);
    if (s->dx2 == 256 && s->kernel == 5) return launch_mel_decoder_256_5(p, grid, st);
    return launch_mel_decoder_256_3(p, grid, st);
}

int esmi_mel_decoder_f32(const float* blob, const esmi_decoder_shape* s, const float* x, const float* h0,
                         const int32_t* cum, const int32_t* mel_len, const int32_t* lmax_dev, int lmax_host, int apply_mask, int B, int T,
                         int L_out, float* mel, void* workspace, size_t workspace_bytes, esmi_stream_t stream) {
    return mel_decoder_launch(blob, s, x, h0, cum, mel_len, lmax_dev, lmax_host, apply_mask, B, T, L_out, mel, workspace, workspace_bytes, stream);
}
}  // extern "C"
