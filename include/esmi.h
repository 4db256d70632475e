/*
 * esmi.h -- C ABI of the MI355X-native EfficientSpeech acoustic-model forward path.
 *
 * The reference (roatienza/efficientspeech) has no FFI: its boundary for this path is the
 * Python nn.Module API of layers/networks.py + layers/blocks.py.  This header is the C-ABI that
 * sits UNDER that API: one entry point per reference stage, plain device pointers and sizes,
 * no torch types.  efficientspeech_amd/networks.py (the host-side mirror of the reference
 * modules) binds these with ctypes; INTEGRATION.md shows the stub a reference maintainer adds.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (hipMalloc'd / torch CUDA tensor .data_ptr()); tensors are
 *    channels-last (B, N, C) contiguous fp32 unless noted; masks are uint8 (1 = padding), the
 *    convention of utils/tools.py:43-51 get_mask_from_lengths.
 *  - functions only ENQUEUE work on `stream` (a hipStream_t); they never allocate, free or
 *    synchronise, so they can be captured into a hipGraph.  Scratch comes from the caller.
 *  - return 0 on success, a negative ESMI_ERR_* for argument errors, a positive hipError_t
 *    if a launch failed.
 *  - conv weights are taken TAP-MAJOR (k, Cout, Cin); use esmi_pack_conv_weight_f32 /
 *    esmi_pack_convT_weight_f32 once at checkpoint-load time to convert from the checkpoint
 *    layouts (Cout, Cin, k) / (Cin, Cout, k).  nn.Linear weights (Cout, Cin) are used as stored.
 *  - the fused (wave-chain) encoder-side kernels additionally take every GEMM weight in MFMA
 *    B-fragment order (the `*_wp` fields, made once with esmi_pack_bfrag_f32): one operand fetch of a
 *    wave is then a single coalesced 1 KiB load.  A NULL `*_wp` selects the one-kernel-per-op plan.
 */
#ifndef ESMI_H
#define ESMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESMI_VERSION 501 /* 0.5.1: esmi_train_conv_ln_fwd_f32 (an addition; nothing else changed shape).  0.5.0 (round 5): esmi_mel_decoder_clock_probe;
                            0.4.0: esmi_decoder_head.proj_w (the decoder's first stage at phoneme rate for every model size:
                          * esmi_decoder_head_f32).  0.3.0: training entry points changed shape (esmi_conv_desc: act / packed_fwd / packed_grad; LayerNorm with
                          * residual / row mask / activation arguments; esmi_train_loss_args.grad_seed; esmi_train_pack_weights_f32,
                          * esmi_train_cat_f32, esmi_reduce_queue); activation-range flag.  0.2.0: launch plan per call (no
                          * process-global state), module-level entry points, weight range guard */

#define ESMI_OK 0
#define ESMI_ERR_ARG (-1)         /* null pointer / bad size */
#define ESMI_ERR_UNSUPPORTED (-2) /* shape outside what the kernels are built for */
#define ESMI_ERR_WORKSPACE (-3)   /* workspace too small */
#define ESMI_ERR_RANGE (-4)       /* range-checked build only: an activation left the binary16 range of the split contractions */

typedef void* esmi_stream_t; /* hipStream_t */

int esmi_version(void);
/* "hip:gfx950" for the product build; "wavesim" for the CPU test simulator build. */
const char* esmi_backend(void);
/* Build-time choices that change HOW (not what) the library computes, e.g. "dec_gemm=split-f16x2,enc_gemm=split-f16x2" (weight
 * GEMMs as fp32-accurate split products on the f16 matrix pipe) or "dec_gemm=fp32-mfma,enc_gemm=fp32-mfma". */
const char* esmi_build_config(void);

/* Launch plan: a bit mask passed PER CALL (esmi_encoder_block_shape.plan, the `plan` argument of
 * esmi_fuse_variance_adaptor_f32) -- the library keeps no mutable state, calls are re-entrant.  A cleared bit forces one kernel
 * per reference op for that stage instead of the fused wave-chain kernel; every plan is parity-tested.  Normal callers pass
 * ESMI_FUSE_ALL. */
#define ESMI_FUSE_MERGE_QKV 1 /* merge convs + 1x1 + qkv                 */
#define ESMI_FUSE_ATTN_FFN 2  /* attention + proj + LN1 + MixFFN + LN2   */
#define ESMI_FUSE_VARIANCE 4  /* Fuse + 3 predictors + embeddings + round */
#define ESMI_FUSE_SPLIT2 8    /* two-head blocks on short sequences: two waves per row tile */
#define ESMI_FUSE_BLOCK 16    /* whole encoder block in one launch when one workgroup covers the sequence */
#define ESMI_FUSE_CHAIN16 32  /* rounds 5-6: the 16-row-tile kernels (weights once per workgroup through LDS) for the shapes they are built
                               * for, one workgroup per utterance -- dim = 32 models: the whole encoder side as chain16 kernels; dim = 64
                               * (T <= 256): enc_va64 / enc_post_attn64; dim = 128 (T <= 256): enc_pred128 / enc_fuse128 / enc_post_attn128 /
                               * enc_merge_q256 (activations in registers end to end).  Without the bit the round-1..4 chain kernels and
                               * the one-kernel-per-op launches run those shapes too */
#define ESMI_FUSE_ALL 63

/* ------------------------------------------------------------------ weight packing
 * nn.Conv1d weight (Cout, Cin, k) -> (k, Cout, Cin)            [networks.py:40-42, blocks.py:17] */
int esmi_pack_conv_weight_f32(const float* src, float* dst, int cout, int cin, int k, esmi_stream_t stream);
/* nn.ConvTranspose1d weight (Cin, Cout, k) -> (k, Cout, Cin)   [networks.py:183] */
int esmi_pack_convT_weight_f32(const float* src, float* dst, int cin, int cout, int k, esmi_stream_t stream);

/* MFMA B-fragment order of `taps` row-major (n, k) matrices (k a multiple of 8; of 32 in the split-f16 build, else
 * ESMI_ERR_UNSUPPORTED), NT = ceil(n/32):
 *   dst[(((t*(k/8) + kc)*NT + nt)*64 + lane)*4 + s] = src[t][32nt + (lane&31)][8kc + 4(lane>>5) + s]   (0 for rows >= n)
 * src is an nn.Linear weight (taps = 1) or a tap-major conv weight (k_taps, Cout, Cin).
 * That is the fp32-MFMA build (enc_gemm=fp32-mfma in esmi_build_config()).  The default build (enc_gemm=split-f16x2)
 * stores the same number of bytes as pre-split binary16 pieces of 2^8 * W: within each group of four kc (32 channels)
 * slot kc&3 = (16-channel step st = (kc>>1)&1, piece pl = kc&1), and dword s of a lane holds the two pieces for channels
 * 32(kc>>2) + 16st + 4(lane>>5) + {2s, 2s+1} (s < 2) or + 8 + {2(s-2), 2(s-2)+1} (s >= 2), low half first
 * (piece 1 = round-to-nearest binary16, piece 2 = binary16 of the remainder; efficientspeech_amd/csrc/small_kernels.h).
 * The blob is opaque to callers either way: only kernels of the same library read it.
 * dst holds esmi_pack_bfrag_floats(n, k, taps) = taps * k * 32 * NT floats.                     */
size_t esmi_pack_bfrag_floats(int n, int k, int taps);
int esmi_pack_bfrag_f32(const float* src, float* dst, int n, int k, int taps, esmi_stream_t stream);

/* The two merge convolutions of an encoder block (networks.py:64-67) are linear and bias-free with nothing in
 * between, so merge1(merge(x)) is ONE k-tap convolution Cin -> Cout with W'[j] = merge1_w @ merge_w[j].
 * merge_w (k,Cin,Cin) tap-major, merge1_w (Cout,Cin) -> dst (k,Cout,Cin) tap-major, accumulated in fp64.   */
int esmi_compose_merge_f32(const float* merge_w, const float* merge1_w, int k, int cin, int cout, float* dst,
                           esmi_stream_t stream);

/* ------------------------------------------------------------------ Encoder block
 * One pass of the loop body of Encoder.forward, layers/networks.py:62-85:
 *   merge convs (dense Conv1d k/stride, then 1x1, both bias-free) -> SelfAttention
 *   (blocks.py:43-71: every head uses the full width C, scale (C//h)^-1/2, scores NOT masked)
 *   -> x = LN1(attn + x) -> mask -> x = LN2(MixFFN(x) + x) -> mask.
 * Block 0 fuses the nn.Embedding gather (networks.py:54): pass ids+embed, x_in = NULL.       */
typedef struct esmi_encoder_block_weights {
    const float* merge_w;  /* (k, Cin, Cin) tap-major  <- attn_blocks.{i}.0.weight (Cin,Cin,k) */
    const float* merge1_w; /* (Cout, Cin)              <- attn_blocks.{i}.1.weight (Cout,Cin,1) */
    const float* qkv_w;    /* (3*h*C, C)               <- attn_blocks.{i}.2.qkv.weight          */
    const float* proj_w;   /* (C, h*C)                 <- attn_blocks.{i}.2.proj.weight         */
    const float* proj_b;   /* (C)                                                               */
    const float* mlp1_w;   /* (e*C, C)                 <- attn_blocks.{i}.3.mlp1.weight         */
    const float* mlp1_b;
    const float* conv_w;   /* (3, e*C, e*C) tap-major  <- attn_blocks.{i}.3.conv.weight         */
    const float* conv_b;
    const float* mlp2_w;   /* (C, e*C)                 <- attn_blocks.{i}.3.mlp2.weight         */
    const float* mlp2_b;
    const float* ln1_g;    /* attn_blocks.{i}.4.{weight,bias} */
    const float* ln1_b;
    const float* ln2_g;    /* attn_blocks.{i}.5.{weight,bias} */
    const float* ln2_b;
    /* MFMA B-fragment copies for the fused kernels (all non-NULL enables them, any NULL -> one kernel per op):
     * merge_cwp = esmi_pack_bfrag_f32(esmi_compose_merge_f32(merge_w, merge1_w), taps = kernel); the others are
     * esmi_pack_bfrag_f32 of the matrix of the same name (conv_w with taps = 3).                              */
    const float* merge_cwp;
    const float* qkv_wp;
    const float* proj_wp;
    const float* mlp1_wp;
    const float* conv_wp;
    const float* mlp2_wp;
    /* Weight-folded attention (optional; all four or none): used by the one-kernel-per-op plan for blocks with two or more heads and,
     * through the *_wp copies, by the whole-block chain kernels for every block.  Every head spans the full width C, so
     *   scores_h = (x Wq_h^T)(x Wk_h^T)^T = (x M_h) x^T          with  M_h = Wq_h^T Wk_h   (C x C)
     *   out      = sum_h softmax(scores_h) (x Wv_h^T) Wp_h^T + b = sum_h (P_h x) O_h + b   with  O_h = Wv_h^T Wp_h^T (C x C):
     * ONE projection per head in front of the attention (q_h = x M_h; keys and values are x itself, shared by all heads) and one
     * behind it instead of q, k, v and proj -- half the projection FLOPs and a third of the activation traffic (3hC -> hC floats per
     * position).  qk_w (h*C, C): rows [h][:] = M_h^T as a Linear weight; vo_w (C, h*C): columns [h] = O_h^T; *_wp = esmi_pack_bfrag_f32.
     * Same function up to fp32 rounding of the folded matrices (computed in fp64 by the caller).                                    */
    const float* qk_w;
    const float* qk_wp;
    const float* vo_w;
    const float* vo_wp;
    /* Block 0 only (optional): the embedding folded into the composed merge convolution.  Embedding, merge conv and merge 1x1 are
     * linear with nothing in between (networks.py:56-67), so x[t] = sum_j E_j[id[t*stride + j - pad]] with
     * E_j = embed @ W'[j]^T, (k, vocab, Cout) row-major, computed once per checkpoint in fp64 by the caller: k row gathers of Cout
     * floats per position instead of a Cin-wide gather and a k*Cin x Cout contraction.  Taps outside [0, n_in) contribute 0 (the
     * conv's zero padding).  Used by the chain kernels (merge_cwp path) when ids are given; NULL = contraction as before.        */
    const float* emb_conv;
    /* MixFFN's Linear(C, e*C) folded into the dense k = 3 conv behind it (blocks.py:22-27: nothing in between), optional, all five or
     * none: ffn_cw[j] = conv_w[j] @ mlp1_w, (3, e*C, C) tap-major, ffn_cwp = esmi_pack_bfrag_f32(ffn_cw, taps = 3) -- a k = 3 conv
     * C -> e*C instead of a Linear and a k = 3 conv e*C -> e*C (expansion 2: 43 % of the two contractions' FLOPs, and the hidden tensor
     * of the Linear is never written).  The Linear's bias reaches every position through the taps that lie inside the sequence:
     * ffn_cb = conv_b + sum_j conv_w[j] @ mlp1_b is the bias of an interior position, the first position lacks
     * ffn_cb_first = conv_w[0] @ mlp1_b and the last one ffn_cb_last = conv_w[2] @ mlp1_b (a sequence of one position lacks both).
     * Computed in fp64 by the caller.  The chain kernels need them (with the other *_wp); the one-kernel-per-op plan uses them when
     * given.  esmi_mixffn_f32 (the module-level MixFFN.forward) keeps the reference's three contractions.                        */
    const float* ffn_cw;
    const float* ffn_cwp;
    const float* ffn_cb;
    const float* ffn_cb_first;
    const float* ffn_cb_last;
} esmi_encoder_block_weights;

typedef struct esmi_encoder_block_shape {
    int B, n_in, c_in;       /* input  (B, n_in, c_in)  (block 0: c_in = embed_dim, n_in = T) */
    int c_out, heads;        /* output (B, n_out, c_out), n_out = (n_in + 2*(k/2) - k)/stride + 1 */
    int kernel, stride;      /* merge conv kernel / stride (networks.py:27-30)                */
    int expansion;           /* MixFFN hidden = expansion * c_out                             */
    int vocab;               /* rows of the embedding table (block 0 only)                    */
    int mask_pool, mask_len; /* `mask` is (B, mask_len) and output row n is padding iff any of
                              * mask[n*mask_pool .. +mask_pool) is set or lies beyond mask_len
                              * (blocks.py:51-57 applied on the fly).  0 / 0 = mask is (B, n_out). */
    int plan;                /* ESMI_FUSE_* bits (launch plan of THIS call)                   */
} esmi_encoder_block_shape;

size_t esmi_encoder_block_workspace_bytes(const esmi_encoder_block_shape* s);
int esmi_encoder_block_f32(const esmi_encoder_block_weights* w, const esmi_encoder_block_shape* s,
                           const int32_t* ids, const float* embed, /* block 0: (B,n_in) ids, (vocab,c_in) table */
                           const float* x_in,                      /* blocks >= 1: (B, n_in, c_in), else NULL   */
                           const uint8_t* mask,                    /* padding mask (see mask_pool) or NULL      */
                           float* x_out,                           /* (B, n_out, c_out)                         */
                           void* workspace, size_t workspace_bytes, esmi_stream_t stream);

/* Pooled padding mask of an encoder block, blocks.py:51-57: pad (B,T) with True to a multiple
 * of `pool`, max over groups of `pool` -> (B, n_out).                                        */
int esmi_pool_mask_u8(const uint8_t* mask, int B, int T, int pool, uint8_t* out, int n_out, esmi_stream_t stream);

/* ------------------------------------------------------------------ Fuse
 * Fuse.forward, layers/networks.py:189-219.  feats[i] = encoder block i output (B, n_i, dim*2^i).
 * out rows are written with leading dimension ld_out (>= dim) so that the result can land
 * directly in the first `dim` channels of the (B,T,4*dim) variance-adaptor tensor.           */
#define ESMI_MAX_DEPTH 4
typedef struct esmi_fuse_weights {
    const float* mlp_w[ESMI_MAX_DEPTH];   /* (dim, dim*2^i)      <- fuse.mlps.{i}.0.weight */
    const float* mlp_b[ESMI_MAX_DEPTH];
    const float* up_w[ESMI_MAX_DEPTH];    /* (k, dim, dim) tap-major <- fuse.mlps.{i>=1}.1.weight (Cin,Cout,k) */
    const float* up_b[ESMI_MAX_DEPTH];
    const float* fuse_w;                  /* (dim, depth*dim)    <- fuse.fuse.weight       */
    const float* fuse_b;
    const float* mlp_wp[ESMI_MAX_DEPTH];  /* esmi_pack_bfrag_f32 of mlp_w / up_w (taps = kernel) / fuse_w; */
    const float* up_wp[ESMI_MAX_DEPTH];   /* used by esmi_fuse_variance_adaptor_f32's fused kernel,        */
    const float* fuse_wp;                 /* NULL -> unfused plan                                          */
} esmi_fuse_weights;
size_t esmi_fuse_workspace_bytes(int B, int T, int dim, int depth);
int esmi_fuse_f32(const esmi_fuse_weights* w, int depth, int dim, int kernel, int B, int T,
                  const float* const* feats, const int* n_i, /* host arrays of `depth` entries */
                  const uint8_t* mask,                       /* (B,T) or NULL                 */
                  float* out, int ld_out, void* workspace, size_t workspace_bytes, esmi_stream_t stream);

/* ------------------------------------------------------------------ Variance adaptor
 * PhonemeEncoder.forward lines 349-384 (layers/networks.py): the three AcousticDecoders
 * (networks.py:151-165; `linear` reads the PRE-norm2 tensor), bucketize + embedding of
 * pitch / energy (networks.py:128-149), the channel concat and the duration rounding
 * (torch.round = half-to-even; masked_fill(mask,0).clamp(min=0) when a mask is given).
 *
 * feat is the (B,T,4*dim) tensor whose first `dim` channels already hold the fused features
 * (esmi_fuse_f32 with ld_out = 4*dim); this call fills channels [dim, 4*dim).                */
typedef struct esmi_predictor_weights {
    const float* conv1_w; /* (3, dim, dim) tap-major <- {x}_decoder.conv1.0.weight */
    const float* conv1_b;
    const float* ln1_g;
    const float* ln1_b;
    const float* conv2_w; /* (3, dim, dim) tap-major */
    const float* conv2_b;
    const float* ln2_g;   /* used by the duration predictor only (features = LN2(y)) */
    const float* ln2_b;
    const float* lin_w;   /* (dim)  <- {x}_decoder.linear.weight (1,dim) */
    const float* lin_b;   /* (1) */
    const float* bins;    /* (dim-1) bucket edges  (pitch / energy), NULL for duration */
    const float* emb;     /* (dim, dim) embedding  (pitch / energy), NULL for duration */
    const float* conv1_wp; /* esmi_pack_bfrag_f32 of conv1_w / conv2_w (taps = 3); NULL -> unfused plan */
    const float* conv2_wp;
} esmi_predictor_weights;
size_t esmi_variance_adaptor_workspace_bytes(int B, int T, int dim);
int esmi_variance_adaptor_f32(const esmi_predictor_weights* pitch, const esmi_predictor_weights* energy,
                              const esmi_predictor_weights* duration, int dim, int B, int T,
                              const uint8_t* mask,          /* (B,T) or NULL (B == 1 path)              */
                              const float* pitch_target,    /* (B,T) teacher values or NULL (use pred)  */
                              const float* energy_target,   /* (B,T) or NULL                            */
                              const int32_t* duration_target, /* (B,T) forced durations or NULL         */
                              float* feat,                  /* (B,T,4*dim) in/out                       */
                              float* pitch_pred, float* energy_pred, float* duration_pred, /* (B,T)     */
                              int32_t* pitch_idx, int32_t* energy_idx, /* (B,T) bucket ids (may be NULL) */
                              int32_t* dur,                 /* (B,T) integer repeat counts              */
                              void* workspace, size_t workspace_bytes, esmi_stream_t stream);

/* Fuse + variance adaptor in ONE call (what PhonemeEncoder.forward does between the encoder and the length
 * regulator, networks.py:347-384); uses a single fused kernel when the shape allows (dim 32 or 64), else the two
 * calls above.  `feat` (B,T,4*dim) is fully written.  Arguments as in esmi_fuse_f32 / esmi_variance_adaptor_f32.
 * cum / mel_len (both or neither): also run the length regulator's scan (esmi_length_regulate_i32 without lmax) --
 * inside the fused kernel when one workgroup covers an utterance (T <= 128), as one more launch otherwise.      */
/* MelDecoder's first stage, proj = Linear(4*dim, dx2) + Tanh + LayerNorm (networks.py:272-276,292), is row-wise and every
 * frame of a phoneme reads the same input row: the fused kernel can compute h0 = LN(tanh(proj(feat))) at PHONEME
 * rate while the features are still on the CU (esmi_mel_decoder_f32 then gathers h0 instead of running the GEMM).   */
typedef struct esmi_decoder_head {
    const float* proj_wp;   /* esmi_pack_bfrag_f32 of decoder.proj.0.weight (dx2, 4*dim) */
    const float* proj_b;    /* decoder.proj.0.bias */
    const float* ln_g;      /* decoder.proj.2.{weight,bias} */
    const float* ln_b;
    int d4, dx2;
    const float* proj_w;    /* decoder.proj.0.weight itself (dx2, 4*dim), or NULL: lets shapes the fused kernel does not serve (small /
                             * base ES, long sequences) run the stage as ONE phoneme-rate GEMM launch (esmi_decoder_head_f32)           */
} esmi_decoder_head;

/* h0 = LayerNorm(tanh(Linear(4*dim, dx2)(feat))) -- MelDecoder's first stage (layers/networks.py:291-293: proj = Linear + Tanh,
 * then LayerNorm) on `rows` rows of phoneme-rate features (rows, 4*dim) -> (rows, dx2).  One GEMM launch with the activation and the
 * LayerNorm in its epilogue; dx2 must be 32, 64, 128 or 256 (one output row per wave).  Needs head->proj_w.                       */
int esmi_decoder_head_f32(const esmi_decoder_head* head, long rows, const float* feat, float* h0, esmi_stream_t stream);

size_t esmi_fuse_variance_adaptor_workspace_bytes(int B, int T, int dim, int depth);
int esmi_fuse_variance_adaptor_f32(const esmi_fuse_weights* fuse, int depth, int dim, int kernel, int B, int T,
                                   const float* const* feats, const int* n_i, const esmi_predictor_weights* pitch,
                                   const esmi_predictor_weights* energy, const esmi_predictor_weights* duration,
                                   const uint8_t* mask, const float* pitch_target, const float* energy_target,
                                   const int32_t* duration_target, float* feat, float* pitch_pred, float* energy_pred,
                                   float* duration_pred, int32_t* pitch_idx, int32_t* energy_idx, int32_t* dur,
                                   int32_t* cum, int32_t* mel_len, /* (B,T), (B) or NULL, NULL                  */
                                   const esmi_decoder_head* head, float* h0, /* (B,T,dx2) or NULL, NULL: computed inside
                                      the fused kernel when it serves the shape (dim 32), else by one more launch
                                      (esmi_decoder_head_f32; needs head->proj_w -- without it ESMI_ERR_UNSUPPORTED,
                                      nothing launched: call again without h0)                                     */
                                   int plan,                                 /* ESMI_FUSE_* bits of this call     */
                                   void* workspace, size_t workspace_bytes, esmi_stream_t stream);

/* ------------------------------------------------------------------ module-level forwards
 * The reference's sub-modules called on their own (one kernel per reference op, fp32 MFMA):
 *  SelfAttention.forward, layers/blocks.py:43-71: qkv Linear (bias-free) -> per-head softmax((q k^T) (C//h)^-1/2) v with every
 *  head at full width C, scores NOT masked -> proj Linear.  x, out: (B,N,C); any N (key-chunked kernel beyond 256 keys). */
size_t esmi_self_attention_workspace_bytes(int B, int N, int C, int heads);
int esmi_self_attention_f32(const float* qkv_w /* (3hC, C) */, const float* proj_w /* (C, hC) */, const float* proj_b, int B, int N,
                            int C, int heads, const float* x, float* out, void* workspace, size_t workspace_bytes,
                            esmi_stream_t stream);
/* MixFFN.forward, layers/blocks.py:22-29: Linear(C, eC) -> dense Conv1d(eC, eC, 3, pad 1) -> exact-erf GELU -> Linear(eC, C).
 * conv_w tap-major (3, eC, eC). */
size_t esmi_mixffn_workspace_bytes(int B, int N, int C, int expansion);
int esmi_mixffn_f32(const float* mlp1_w, const float* mlp1_b, const float* conv_w, const float* conv_b, const float* mlp2_w,
                    const float* mlp2_b, int B, int N, int C, int expansion, const float* x, float* out, void* workspace,
                    size_t workspace_bytes, esmi_stream_t stream);
/* AcousticDecoder.forward, layers/networks.py:151-165: conv1+ReLU -> LN1 -> ReLU -> conv2+ReLU; pred = Linear(dim,1) of the
 * PRE-norm2 tensor (ReLU'd for the duration predictor, which also returns features = LN2(.)).  x rows have leading dimension
 * ldx >= dim.  workspace: esmi_variance_adaptor_workspace_bytes(B, T, dim). */
int esmi_acoustic_decoder_f32(const esmi_predictor_weights* w, int dim, int B, int T, int duration, const float* x, int ldx,
                              float* pred /* (B,T) */, float* features /* (B,T,dim), duration predictor only, else NULL */,
                              void* workspace, size_t workspace_bytes, esmi_stream_t stream);
/* AcousticDecoder.get_embedding, layers/networks.py:128-149: idx = torch.bucketize(v, bins) (right=False, dim-1 edges),
 * out[row] = emb[idx] (dim floats); idx (rows) may be NULL. */
int esmi_bucket_embedding_f32(const float* v, const float* bins, const float* emb, int64_t rows, int dim, float* out,
                              int32_t* idx, esmi_stream_t stream);

/* ------------------------------------------------------------------ operand-range guard of the split-f16 build
 * The default build computes weight GEMMs as fp32-accurate split products on the f16 matrix pipe (HISTORY.md 3): 2^8 * W must
 * be a finite binary16 number.  esmi_split_weight_limit() is the largest admissible |W| (inf for the fp32-MFMA build);
 * esmi_absmax_f32 reduces max|x| (NaN -> inf) into a device float so that a loader can check a checkpoint ONCE at pack time
 * (efficientspeech_amd/networks.py raises ValueError and names libesmi_fp32mfma.so as the build to use instead).
 * Activations saturate gracefully: the first piece is converted round-toward-zero (never inf), so values up to 131008 are
 * represented by the two pieces (with fewer bits above 65504); LayerNorm / tanh / GELU outputs are orders of magnitude below. */
float esmi_split_weight_limit(void);
int esmi_absmax_f32(const float* x, int64_t n, float* out, esmi_stream_t stream);

/* ------------------------------------------------------------------ Length regulator
 * FeatureUpsampler.forward, layers/networks.py:228-258 (and its dead twin acoustic.py:33-42):
 * frame j of utterance b is phoneme i with cum[b][i-1] <= j < cum[b][i], cum = inclusive
 * cumsum of max(dur,0).  Writes cum (B,T), mel_len (B) and *lmax = max_b mel_len[b]
 * (all on device: no host round-trip, unlike the reference's 2*B syncs).                     */
int esmi_length_regulate_i32(const int32_t* dur, int B, int T, int32_t* cum, int32_t* mel_len, int32_t* lmax,
                             esmi_stream_t stream);
/* out[0] = max(v[0..n), 0) -- e.g. the padded length lmax from mel_len when the scan ran inside
 * esmi_fuse_variance_adaptor_f32 (one small workgroup, no atomics).                                 */
int esmi_max_i32(const int32_t* v, int n, int32_t* out, esmi_stream_t stream);
/* explicit frame -> phoneme map idx (B,L): -1 for padding frames (j >= mel_len[b]). */
int esmi_length_regulator_indices_i32(const int32_t* cum, int B, int T, int L, int32_t* idx, esmi_stream_t stream);
/* materialise `features` (B,L,C) and `masks` (B,L) u8 exactly as FeatureUpsampler returns them
 * (padding: 0.0 / True).  fmask: (B,T) phoneme mask or NULL.                                  */
int esmi_upsample_f32(const float* feat, const uint8_t* fmask, const int32_t* cum, int B, int T, int C, int L,
                      float* features, uint8_t* masks, esmi_stream_t stream);

/* ------------------------------------------------------------------ Mel decoder
 * MelDecoder.forward, layers/networks.py:291-304, fully fused: proj Linear+Tanh+LN, n_blocks x
 * [block_depth x (depthwise k conv -> pointwise conv -> Tanh -> LN); skip LN], mel Linear.
 * Weights are packed once into one blob (MFMA B-fragment order).  In the default build the
 * matrices are stored as two binary16 pieces of 2^8 * W (same bytes as fp32; esmi_build_config() says
 * "dec_gemm=split-f16x2"): |W| must be < 255 and activations inside the binary16 range (HISTORY.md 3);
 * the blob is opaque and only valid for the library that packed it.                                  */
#define ESMI_MAX_DEC_LAYERS 16
typedef struct esmi_decoder_weights {   /* checkpoint layouts, device pointers */
    const float* proj_w;   /* (dx2, d4)     <- decoder.proj.0.weight */
    const float* proj_b;
    const float* proj_ln_g; /* decoder.proj.2.{weight,bias} */
    const float* proj_ln_b;
    const float* dw_w[ESMI_MAX_DEC_LAYERS]; /* (dx2,1,k)   <- decoder.blocks.{b}.0.{d}.0.0.weight */
    const float* dw_b[ESMI_MAX_DEC_LAYERS];
    const float* pw_w[ESMI_MAX_DEC_LAYERS]; /* (dx2,dx2,1) <- decoder.blocks.{b}.0.{d}.0.1.weight */
    const float* pw_b[ESMI_MAX_DEC_LAYERS];
    const float* ln_g[ESMI_MAX_DEC_LAYERS]; /* decoder.blocks.{b}.0.{d}.1.{weight,bias}           */
    const float* ln_b[ESMI_MAX_DEC_LAYERS];
    const float* skip_g[ESMI_MAX_DEC_LAYERS]; /* decoder.blocks.{b}.1.{weight,bias}               */
    const float* skip_b[ESMI_MAX_DEC_LAYERS];
    const float* mel_w;    /* (n_mel, dx2)  <- decoder.mel_linear.weight */
    const float* mel_b;
} esmi_decoder_weights;

typedef struct esmi_decoder_shape {
    int d4, dx2;              /* input channels 4*dim, hidden min(4*dim, 256): dx2 in {128, 256} */
    int kernel;               /* depthwise kernel (3 or 5)                                       */
    int n_blocks, block_depth;
    int n_mel;                /* <= 96                                                           */
} esmi_decoder_shape;

size_t esmi_mel_decoder_blob_bytes(const esmi_decoder_shape* s);
int esmi_mel_decoder_pack_f32(const esmi_decoder_weights* w, const esmi_decoder_shape* s, float* blob,
                              esmi_stream_t stream);

/* Two input modes:
 *  - fused length-regulator (cum != NULL): `x` is the phoneme-rate tensor (B,T,d4); frame j of
 *    utterance b reads row searchsorted(cum[b], j); frames >= mel_len[b] read zeros.  This is
 *    Phoneme2Mel.forward's encoder->decoder hand-off without materialising (B,L,d4).
 *  - direct (cum == NULL): `x` is (B,L,d4) exactly as MelDecoder.forward receives it.
 * L is the padded length the reference's Conv1d sees (zero padding beyond it): *lmax_dev if
 * lmax_dev != NULL, else lmax_host if >= 0, else max_b mel_len[b] (every workgroup derives it from
 * mel_len, which is then required).  mel is (B, L_out, n_mel); rows in [L, L_out) are zeroed.
 * mel_len != NULL && apply_mask: rows >= mel_len[b] are zeroed (Phoneme2Mel's final
 * masked_fill, networks.py:424-427).                                                          */
int esmi_mel_decoder_f32(const float* blob, const esmi_decoder_shape* s, const float* x,
                         const float* h0, /* optional, fused mode only: (B,T,dx2) = LN(tanh(proj(x))), see esmi_decoder_head */
                         const int32_t* cum, const int32_t* mel_len, const int32_t* lmax_dev, int lmax_host, int apply_mask, int B,
                         int T, int L_out, float* mel,
                         void* workspace, size_t workspace_bytes, /* esmi_mel_decoder_workspace_bytes, or NULL / 0: dx2 = 256 then
                                                                     recomputes both halos of every 128-frame window (slower) */
                         esmi_stream_t stream);
/* scratch for the dx2 = 256 kernel's carried rows (a workgroup walks its share of an utterance chunk by chunk; 0 for dx2 = 128).
 * L_out < 0: an upper bound for ANY output length (for callers that size their scratch before the length is known).             */
size_t esmi_mel_decoder_workspace_bytes(const esmi_decoder_shape* s, int B, int L_out);
/* Measurement aid (bench.py `roofline.clock`), not part of the data path: arm (dev_slots != NULL: 4 int64 in device memory) or disarm
 * (NULL) a probe that the FIRST workgroup of every following esmi_mel_decoder_f32 launch fills with {shader clock, 100 MHz clock} read
 * when it starts (slots 0, 1) and when it starts its last chunk / its last stage (slots 2, 3): (shader ticks / 100 MHz ticks) x 100 MHz
 * is the clock the CU ran the kernel at, which `roofline.peak` (quoted at 2.4 GHz) has to be scaled by.
 * Contract: the ONE piece of state the library keeps (a device global per decoder translation unit).  It is set on the CURRENT device
 * only (single-device processes; a multi-GPU process arms it per device under hipSetDevice), it is process-wide (not per stream), and
 * the caller MUST disarm it (NULL) before freeing the slots -- an armed probe makes every decoder launch store through the pointer.  */
int esmi_mel_decoder_clock_probe(int64_t* dev_slots);
/* ------------------------------------------------------------------ whole inference forward in ONE call
 * Phoneme2Mel.forward (eval), layers/networks.py:415-434 = Encoder blocks -> Fuse + variance adaptor (+ length-regulator scan,
 * + the decoder's phoneme-rate first stage when the fused kernel serves the shape) -> fused mel decoder, enqueued by a C
 * host loop: one FFI crossing per forward instead of one per stage (the Python glue between the per-stage calls cost
 * ~0.1 ms per forward in round 1 -- more than the GPU time of a 32-utterance shard).  Same kernels, same results as the
 * per-stage entry points above.
 *
 * Scratch: ONE caller-provided arena of esmi_forward_arena_bytes() bytes (16-byte aligned base); nothing in it needs to
 * survive the call.  Outputs the reference returns are separate caller buffers: mel, mel_len, duration_pred.  When L_out > 0 at
 * sizing time the arena's scratch also covers esmi_mel_decoder_workspace_bytes(dec_shape, B, L_out) (the dx2 = 256 decoder's carried
 * rows); with L_out unknown then (length taken from the device) a long batch may run the decoder's window form instead.       */
typedef struct esmi_forward_args {
    int B, T, depth, dim, fuse_kernel, plan;
    esmi_encoder_block_weights blocks[ESMI_MAX_DEPTH];
    esmi_encoder_block_shape shapes[ESMI_MAX_DEPTH]; /* B / n_in / mask_pool / mask_len / plan are filled in by the call */
    const float* embed;
    esmi_fuse_weights fuse;
    esmi_predictor_weights pitch, energy, duration;
    esmi_decoder_head head;       /* proj_wp == NULL: the decoder runs its first stage itself                     */
    const float* dec_blob;
    esmi_decoder_shape dec_shape;
    /* inputs */
    const int32_t* ids;           /* (B,T)                                                                         */
    const uint8_t* mask;          /* (B,T) or NULL (the reference's B == 1 path)                                   */
    const int32_t* dur_forced;    /* (B,T) or NULL: injected durations (extension)                  */
    /* outputs */
    float* duration_pred;         /* (B,T)                                                                         */
    int32_t* mel_len;             /* (B)                                                                           */
    float* mel;                   /* (B, L_out, n_mel)                                                             */
    int L_out;                    /* allocation length of mel; rows >= the batch's padded length are zeroed        */
    int lmax_host;                /* >= 0: the caller knows the padded length exactly; -1: derived from mel_len on the device */
    int32_t* lmax_dev;            /* optional out: device scalar max_b mel_len[b] (multi-GPU MAX-reduce); NULL to skip */
    /* optional taps (NULL to skip): what PhonemeEncoder._encode exposes to tests / training-style callers */
    float* pitch_pred; float* energy_pred; int32_t* pitch_idx; int32_t* energy_idx; int32_t* dur; int32_t* cum;
    void* arena; size_t arena_bytes;
    int32_t* range_flag;   /* NULL (normal operation), or a device word: ask for the activation-range check.  Only the checked build
                            * (libesmi_checked.so, -DESMI_RANGE_CHECK=1) honours it: every value that enters a split-f16 contraction is
                            * tested against the binary16 range (|a| < 65504), the call then WAITS for the stream and returns
                            * ESMI_ERR_RANGE when one was outside -- a validation mode for new checkpoints, not for serving.  The
                            * unchecked libraries return ESMI_ERR_UNSUPPORTED when the field is set */
} esmi_forward_args;
size_t esmi_forward_arena_bytes(const esmi_forward_args* a);
/* stage = 0: everything; 1: encoder side only (up to cum / mel_len / lmax_dev / h0, kept in the arena); 2: mel decoder only,
 * on the arena a stage-1 call filled (same args) -- lets a multi-GPU caller put its MAX all-reduce of lmax_dev in between. */
int esmi_phoneme2mel_forward_f32(const esmi_forward_args* a, int stage, esmi_stream_t stream);

/* ------------------------------------------------------------------ HiFi-GAN generator (the vocoder behind model.py:161-162)
 * hifigan/models.py:84-135 Generator.forward with ResBlock1 (:20-58) or ResBlock2 (:61-82), weights as after
 * remove_weight_norm():  x = conv_pre(mel); per stage i: x = ConvTranspose1d_i(leaky_relu(x, 0.1)), x = mean_j ResBlock_ij(x);
 * wav = tanh(conv_post(leaky_relu(x, 0.01))).  Channels-last throughout: mel (B, L, n_mel) is the acoustic model's own output
 * layout (the reference transposes it to channels-first for its Conv1d; nothing is transposed here), wav (B, L * prod(rates)).
 * Every convolution is one implicit-GEMM launch (convgemm.h) with the leaky_relu applied to the input rows as they are
 * loaded, the residual add / the sum over the ResBlocks of a stage folded into the epilogue and the 1/num_kernels of the mean
 * folded into the next convolution's input scale (leaky_relu is positively homogeneous).
 * All conv weights tap-major (k, Cout, Cin): esmi_pack_conv_weight_f32 / esmi_pack_convT_weight_f32 of the checkpoint tensors. */
#define ESMI_HIFIGAN_MAX_UP 8
#define ESMI_HIFIGAN_MAX_KERNELS 8
#define ESMI_HIFIGAN_MAX_RBCONV 96 /* (resblock index n = stage * n_kernels + j) * 3 + conv index m */
typedef struct esmi_hifigan_weights {
    const float* pre_w;  /* <- conv_pre.weight (C0, n_mel, 7) */
    const float* pre_b;
    const float* up_w[ESMI_HIFIGAN_MAX_UP];  /* <- ups.{i}.weight (Cin, Cout, k) via esmi_pack_convT_weight_f32 */
    const float* up_b[ESMI_HIFIGAN_MAX_UP];
    const float* rb_w1[ESMI_HIFIGAN_MAX_RBCONV]; /* <- resblocks.{n}.convs1.{m} (ResBlock1) / resblocks.{n}.convs.{m} (ResBlock2) */
    const float* rb_b1[ESMI_HIFIGAN_MAX_RBCONV];
    const float* rb_w2[ESMI_HIFIGAN_MAX_RBCONV]; /* <- resblocks.{n}.convs2.{m} (ResBlock1 only) */
    const float* rb_b2[ESMI_HIFIGAN_MAX_RBCONV];
    const float* post_w; /* <- conv_post.weight (1, C_last, 7) */
    const float* post_b;
    /* esmi_pack_resblock_f16 of rb_w1[n] / rb_w2[n] (optional).  When every conv of a ResBlock has one and its channel count
     * is 8, 16, 32 or 64, the block runs as ONE launch on an LDS-resident window (csrc/hifigan_resblock.h) instead of one
     * launch per convolution; NULL -> conv by conv.  (libesmi_fp32mfma.so ignores them.) */
    const void* rb_wp1[ESMI_HIFIGAN_MAX_RBCONV];
    const void* rb_wp2[ESMI_HIFIGAN_MAX_RBCONV];
} esmi_hifigan_weights;
typedef struct esmi_hifigan_shape {
    int n_mel, initial_channel, n_up, n_kernels;
    int resblock;                                  /* 1 or 2 */
    int up_rates[ESMI_HIFIGAN_MAX_UP], up_kernels[ESMI_HIFIGAN_MAX_UP];
    int rb_kernels[ESMI_HIFIGAN_MAX_KERNELS];
    int rb_dilations[ESMI_HIFIGAN_MAX_KERNELS * 3]; /* [j*3 + m] */
} esmi_hifigan_shape;
/* Fragment-ordered split-f16 form of one ResBlock convolution: src = tap-major (k, C, C) fp32 (esmi_pack_conv_weight_f32),
 * dst = esmi_pack_resblock_bytes(c, k) bytes (0: no fused kernel for this shape: c in {8, 16, 32, 64}, k in {3, 7, 11}). */
size_t esmi_pack_resblock_bytes(int c, int k);
int esmi_pack_resblock_f16(const float* src, void* dst, int c, int k, esmi_stream_t stream);
size_t esmi_hifigan_workspace_bytes(const esmi_hifigan_shape* s, int B, int L);
int esmi_hifigan_generator_f32(const esmi_hifigan_weights* w, const esmi_hifigan_shape* s, const float* mel, int B, int L,
                               float* wav, void* workspace, size_t workspace_bytes, esmi_stream_t stream);

/* x.masked_fill(mask[:, :, None], 0) on (rows, C) fp32 -- used by the module-level API when the
 * decoder is called stand-alone.                                                              */
int esmi_mask_rows_f32(float* x, const uint8_t* mask, int64_t rows, int C, esmi_stream_t stream);

/* ------------------------------------------------------------------ Training step
 * model.py:167-226 (loss, training_step), :279-283 (AdamW); the train=True data flow of layers/networks.py:336-434.
 * Operator-level entry points: each forward keeps what its backward reads; activations are channels-last (B, n, C) fp32,
 * weights and their gradients are in CHECKPOINT layout (what the optimizer updates): Conv1d (Cout, Cin/groups, k),
 * ConvTranspose1d (Cin, Cout, k), Linear (Cout, Cin) = a conv with k = 1.  First correct version: one thread per output
 * element, fp32 FMA loops, no atomics (a step is bitwise reproducible); see csrc/train_ops.h.  The Python side
 * (efficientspeech_amd/train.py) composes them with torch.autograd as the tape. */
typedef struct esmi_conv_desc {
    int B, n_in, c_in, n_out, c_out, k, stride, pad;
    int groups;       /* 1, or c_in == c_out == groups (depthwise, MelDecoder networks.py:275) */
    int transposed;   /* 1: nn.ConvTranspose1d (Fuse, networks.py:185), groups must be 1 */
    int precision;    /* 0 / 32: fp32-accurate contractions.  16: the reference's `--precision 16` (utils/tools.py:326-327, Lightning ->
                       * torch.autocast): the forward and data-gradient GEMMs round both operands to binary16 and run ONE MFMA product
                       * per 16 channels (fp32 accumulate, fp32 tensors in memory, fp32 master weights); weight gradients stay fp32 */
    int act;          /* forward only: 0, or 1 ReLU / 2 GELU (erf) / 3 tanh applied to the result (in the GEMM's epilogue when the shape runs
                       * on the matrix pipe, else as a second launch); the backward entry points ignore it -- the caller applies
                       * esmi_train_act_bwd_f32 to dy first (ReLU / tanh from the saved OUTPUT) */
    /* optional (NULL: the GEMM copy of the weight is made per call, in the workspace): persistent copies of THIS step's weight for
     * the forward / the data-gradient problem, esmi_train_conv_workspace_bytes(d) each, filled by esmi_train_pack_weights_f32
     * after the optimizer's last update -- one launch for every convolution of the model instead of one per call */
    float* packed_fwd;
    float* packed_grad;
} esmi_conv_desc;
/* workspace (esmi_train_conv_workspace_bytes, may be NULL): scratch for the tap-major weight copy; with it, dense convolutions
 * run on the matrix pipe through the inference path's implicit GEMM (split-f16 products in libesmi.so, fp32 MFMA in
 * libesmi_fp32mfma.so) -- the data gradient as the transposed problem; without it every shape takes the plain fp32 kernels */
size_t esmi_train_conv_workspace_bytes(const esmi_conv_desc* d);
/* the GEMM copies of n weights in ONE launch (per 64): descs[i].packed_fwd / packed_grad (each may be NULL) <- weights[i] in the layout
 * the forward / data-gradient GEMM of descs[i] reads; also clears the operand-scale slot behind each data-gradient copy.  Only the
 * channel / tap / stride / padding / transposed fields of a descriptor matter here.  Call it once per step, before the forward. */
int esmi_train_pack_weights_f32(const esmi_conv_desc* descs, const float* const* weights, int n, esmi_stream_t stream);
int esmi_train_conv_fwd_f32(const esmi_conv_desc* d, const float* x, const float* w, const float* bias /* or NULL */, float* y,
                            void* workspace, size_t workspace_bytes, esmi_stream_t stream);
int esmi_train_conv_dgrad_f32(const esmi_conv_desc* d, const float* dy, const float* w, float* dx, void* workspace,
                              size_t workspace_bytes, esmi_stream_t stream);
/* weight (and bias) gradients: a two-stage reduction over the B * n_out rows in fixed order (reproducible); scratch from the caller */
size_t esmi_train_conv_wgrad_workspace_bytes(const esmi_conv_desc* d);
int esmi_train_conv_wgrad_f32(const esmi_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias /* or NULL */,
                              void* workspace, size_t workspace_bytes, esmi_stream_t stream);
/* Deferred second stage of the two-stage (fixed-order, reproducible) reductions: parameter gradients are only read by the optimizer at
 * the end of the step, so the callers below may queue their "sum the partial rows" stage instead of launching it; ONE launch
 * (esmi_train_reduce_flush_f32) then runs every queued reduction.  The queue is a caller-owned host struct (the library keeps no
 * state); the partial buffers (workspaces) must stay alive until the flush.  A full queue is flushed by the call that finds it full. */
typedef struct esmi_reduce_item {
    const float* partial; int64_t n, stride, chunks; float* out; int64_t n0; float* out1;   /* out[e] (e < n0 or no out1) / out1[e - n0] = sum_c partial[c * stride + e] */
} esmi_reduce_item;
#define ESMI_REDUCE_QUEUE_ITEMS 48
typedef struct esmi_reduce_queue { int32_t count; esmi_reduce_item items[ESMI_REDUCE_QUEUE_ITEMS]; } esmi_reduce_queue;
int esmi_train_reduce_flush_f32(esmi_reduce_queue* q, esmi_stream_t stream);
/* both gradients of one convolution in one call (what autograd's backward of the op needs): the weight-gradient pass leaves max|dy|
 * behind as a by-product and the data-gradient GEMM takes its operand scale from it -- no separate pass over dy.  Shapes that do not
 * run on the matrix pipe fall back to the two entry points above. */
size_t esmi_train_conv_bwd_workspace_bytes(const esmi_conv_desc* d);
int esmi_train_conv_bwd_f32(const esmi_conv_desc* d, const float* x, const float* dy, const float* w, float* dx, float* dw,
                            float* dbias /* or NULL */, void* workspace, size_t workspace_bytes, esmi_reduce_queue* defer /* or NULL */,
                            esmi_stream_t stream);
/* nn.LayerNorm over the last dim (eps 1e-5); mean / rstd (rows) are kept for the backward.  The steps the reference runs around it
 * ride in the same launch: res (or NULL) -- the normalised tensor is x + res, written to xsum for the backward (networks.py:75,83,301:
 * LN(f(x) + x); the gradient of both summands is the backward's dx); rowmask (or NULL, rows bytes) -- rows with a nonzero byte come
 * out zero (the masked_fill of networks.py:76,84) and pass no gradient: give the backward the same mask. */
int esmi_train_layernorm_fwd_f32(const float* x, const float* g, const float* b, int64_t rows, int C, float* y, float* mean,
                                 float* rstd, const float* res /* or NULL */, float* xsum /* with res */,
                                 const uint8_t* rowmask /* or NULL */, int relu_out /* 1: y = relu(LN(.)), networks.py:153-154 */,
                                 esmi_stream_t stream);
/* A stride-1 convolution / Linear with the LayerNorm behind it in ONE launch (round 5: the norm rides in the GEMM's epilogue, as it does
 * in the inference plans): y_pre = act(conv(x) + bias) + res is what the norm's backward reads as its `x` (with in_act = d->act when
 * the activation's backward is to run there), y = mask(relu_out(LN(y_pre))), mean / rstd per row.  Same arguments as the two calls it
 * replaces (esmi_train_conv_fwd_f32, then esmi_train_layernorm_fwd_f32 with xsum = y_pre).  ESMI_ERR_UNSUPPORTED when the shape does
 * not run as a GEMM with a whole row per wave (c_out not in {32, 64, 128}, strided / transposed, no workspace): make the two
 * calls then -- the results agree to fp32 rounding (the sums of the norm run in a different order), the backward is the same. */
int esmi_train_conv_ln_fwd_f32(const esmi_conv_desc* d, const float* x, const float* w, const float* bias /* or NULL */,
                               const float* res /* or NULL */, const float* ln_g, const float* ln_b,
                               const uint8_t* rowmask /* or NULL */, int relu_out, float* y_pre, float* y, float* mean, float* rstd,
                               void* workspace, size_t workspace_bytes, esmi_stream_t stream);
size_t esmi_train_layernorm_bwd_workspace_bytes(int64_t rows, int C);
int esmi_train_layernorm_bwd_f32(const float* x, const float* g, const float* mean, const float* rstd, const float* dy,
                                 int64_t rows, int C, float* dx, float* dg, float* db, void* workspace, size_t workspace_bytes,
                                 esmi_reduce_queue* defer /* or NULL */, const uint8_t* rowmask /* or NULL */,
                                 int in_act /* 0, or 1 ReLU / 3 tanh: x is that activation's output and dx is returned for its INPUT */,
                                 const float* y_relu /* NULL, or the forward's output when it ran with relu_out: dy counts only where y > 0 */,
                                 esmi_stream_t stream);
/* kind: 1 ReLU, 2 GELU (erf), 3 tanh.  Backward reads the OUTPUT for ReLU / tanh and the INPUT for GELU as `saved`. */
int esmi_train_act_fwd_f32(const float* x, int64_t n, int kind, float* y, esmi_stream_t stream);
int esmi_train_act_bwd_f32(const float* saved, const float* dy, int64_t n, int kind, float* dx, esmi_stream_t stream);
/* attention core of blocks.py:43-64 (scores unmasked, scale (C/h)^-1/2): qkv (B, N, 3, h, C) -> P (B, h, N, N), ctx (B, N, h*C);
 * backward: dqkv from dctx, scratch dS (B, h, N, N) */
int esmi_train_attention_fwd_f32(const float* qkv, int B, int N, int C, int h, float* P, float* ctx, esmi_stream_t stream);
int esmi_train_attention_bwd_f32(const float* qkv, const float* P, const float* dctx, int B, int N, int C, int h, float* dS,
                                 float* dqkv, esmi_stream_t stream);
/* nn.Embedding: out[r] = table[ids[r]]; dtable[v] = sum of dy rows with ids == v (none for v == padding_idx; -1: no padding row) */
int esmi_train_embedding_fwd_f32(const int32_t* ids, const float* table, int64_t rows, int V, int C, float* out, esmi_stream_t stream);
size_t esmi_train_embedding_bwd_workspace_bytes(int64_t rows, int V, int C);
int esmi_train_embedding_bwd_f32(const int32_t* ids, const float* dy, int64_t rows, int V, int C, int padding_idx, float* dtable,
                                 void* workspace, size_t workspace_bytes, esmi_reduce_queue* defer /* or NULL */, esmi_stream_t stream);
int esmi_train_mask_rows_f32(const float* x, const uint8_t* mask, int64_t rows, int C, float* y, esmi_stream_t stream);
/* torch.cat(parts, dim=-1) of n <= 8 row-major (rows, widths[j]) parts into cat (rows, sum widths) in one launch -- backward = 1: the
 * slices of `cat` (then the gradient) back into the parts.  Parts whose bit is set in masked_parts count as zero on rows with
 * rowmask[r] != 0 (the masked_fill in front of the cat, networks.py:366-368), in both directions. */
int esmi_train_cat_f32(float* const* parts, const int* widths, int n, int64_t rows, float* cat, const uint8_t* rowmask /* or NULL */,
                       unsigned masked_parts, int backward, esmi_stream_t stream);
int esmi_train_add_f32(const float* a, const float* b, int64_t n, float* y, esmi_stream_t stream);
/* dst[r, col_dst + c] = src[r, col_src + c], c < C: torch.cat along channels and its gradient */
int esmi_train_copy_cols_f32(const float* src, int ld_src, int col_src, float* dst, int ld_dst, int col_dst, int64_t rows, int C,
                             esmi_stream_t stream);
/* FeatureUpsampler (networks.py:228-258) on the inclusive duration cumsum (esmi_length_regulate_i32): (B, T, C) -> (B, L, C),
 * frames past the utterance zero; backward = segment sums */
int esmi_train_repeat_fwd_f32(const float* feat, const int32_t* cum, int B, int T, int C, int L, float* out, esmi_stream_t stream);
int esmi_train_repeat_bwd_f32(const float* dout, const int32_t* cum, int B, int T, int C, int L, float* dfeat, esmi_stream_t stream);
/* model.py:167-216: masked L1 (mel) + masked MSE (pitch, energy, log(duration + 1)), total = 10 a + 2 b + 2 c + d.
 * out (5 floats, device): the four means and the total; d_*: d total / d prediction (0 under the masks). */
#define ESMI_TRAIN_LOSS_SCRATCH_FLOATS 1536
typedef struct esmi_train_loss_args {
    const float *mel_pred, *mel;            /* (B, L, n_mel) */
    const float *pitch_pred, *pitch, *energy_pred, *energy, *dur_pred; /* (B, T) */
    const int32_t* dur;                     /* (B, T) target repeat counts */
    const uint8_t *mel_mask, *ph_mask;      /* (B, L) / (B, T), 1 = padding; NULL = nothing masked */
    int B, T, L, n_mel;
    float *out, *d_mel, *d_pitch, *d_energy, *d_dur;
    float* scratch;                         /* ESMI_TRAIN_LOSS_SCRATCH_FLOATS floats (partial sums of the two-stage reduction) */
    const float* grad_seed;                 /* NULL, or one device float every d_* is multiplied by (the backward's seed: the loss scale) */
} esmi_train_loss_args;
int esmi_train_loss_f32(const esmi_train_loss_args* a, esmi_stream_t stream);
/* torch.optim.AdamW's update of one flat buffer (model.py:279-283); step >= 1.  The hyper-parameters are doubles, as the Python
 * scalars torch.optim.AdamW holds them: the bias corrections 1 - beta^step are evaluated in double precision on the host and
 * every derived scalar (1 - beta, lr * weight_decay, lr / bc1, sqrt(bc2)) is rounded to fp32 once, like torch's scalar arguments */
int esmi_train_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                         double weight_decay, int step, double grad_scale /* gradients are g * grad_scale: 1 / loss scale under
                         precision 16 (torch.amp.GradScaler's unscale_), else 1 */, esmi_stream_t stream);
/* the same update with its bookkeeping in device memory -- for a captured hipGraph of the whole step, and for `precision=16` without a
 * host round trip: hyper_dev = ESMI_TRAIN_ADAMW_HYPER_FLOATS floats ([0] = learning rate, written by the caller before a replay;
 * the rest is scratch of this call: the bias corrections of the current step, evaluated in double precision by one device thread,
 * the skip flag and the gradient scale), step_dev = the step counter (1 int32, advanced by this call before the update).
 * grad_absmax + scaler_state (both NULL, or both given): torch.amp.GradScaler on the device -- grad_absmax = max|g| of the (scaled)
 * gradient buffer with nan ordered as inf (esmi_absmax_f32), scaler_state = ESMI_TRAIN_SCALER_FLOATS floats {scale, growth_factor,
 * backoff_factor, growth_interval, clean steps in a row, skipped steps}: an overflow skips the update (step_dev stays) and backs
 * the scale off; otherwise the update runs on g / scale and the growth counter advances */
#define ESMI_TRAIN_ADAMW_HYPER_FLOATS 8
#define ESMI_TRAIN_SCALER_FLOATS 8
int esmi_train_adamw_graph_f32(float* p, const float* g, float* m, float* v, int64_t n, float* hyper_dev, double beta1, double beta2,
                               double eps, double weight_decay, int32_t* step_dev, const float* grad_absmax /* or NULL */,
                               float* scaler_state /* or NULL */, esmi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ESMI_H */
