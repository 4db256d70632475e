"""ctypes binding of the esmi C-ABI (include/esmi.h).

The product library is `efficientspeech_amd/libesmi.so`, built for gfx950 by
`__graft_entry__.build()` (hipcc).  There is NO CPU fallback: if the library is missing
or fails to load, `load()` raises.  (tests/ may bind the wave-simulator build of the very
same sources through `bind()`; nothing in this package does.)
"""
import contextlib
import itertools
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ESMI_LIB") or os.path.join(_HERE, "libesmi.so")   # ESMI_LIB: another build of the same ABI
MAX_DEPTH = 4
MAX_DEC_LAYERS = 16

ESMI_OK = 0
_ERRS = {-1: "ESMI_ERR_ARG (null pointer / bad size / misaligned)",
         -2: "ESMI_ERR_UNSUPPORTED (shape outside what the kernels are built for)",
         -3: "ESMI_ERR_WORKSPACE (workspace too small)",
         -4: "ESMI_ERR_RANGE (an activation left the binary16 range of the split contractions)"}

fp = C.c_void_p  # device pointers travel as integers


class EncoderBlockWeights(C.Structure):
    _fields_ = [(n, fp) for n in ("merge_w", "merge1_w", "qkv_w", "proj_w", "proj_b", "mlp1_w", "mlp1_b",
                                  "conv_w", "conv_b", "mlp2_w", "mlp2_b", "ln1_g", "ln1_b", "ln2_g", "ln2_b",
                                  "merge_cwp", "qkv_wp", "proj_wp", "mlp1_wp", "conv_wp", "mlp2_wp",
                                  "qk_w", "qk_wp", "vo_w", "vo_wp", "emb_conv",
                                  "ffn_cw", "ffn_cwp", "ffn_cb", "ffn_cb_first", "ffn_cb_last")]


class EncoderBlockShape(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "n_in", "c_in", "c_out", "heads", "kernel", "stride", "expansion", "vocab",
                                       "mask_pool", "mask_len", "plan")]


class FuseWeights(C.Structure):
    _fields_ = [("mlp_w", fp * MAX_DEPTH), ("mlp_b", fp * MAX_DEPTH), ("up_w", fp * MAX_DEPTH),
                ("up_b", fp * MAX_DEPTH), ("fuse_w", fp), ("fuse_b", fp),
                ("mlp_wp", fp * MAX_DEPTH), ("up_wp", fp * MAX_DEPTH), ("fuse_wp", fp)]


class PredictorWeights(C.Structure):
    _fields_ = [(n, fp) for n in ("conv1_w", "conv1_b", "ln1_g", "ln1_b", "conv2_w", "conv2_b", "ln2_g", "ln2_b",
                                  "lin_w", "lin_b", "bins", "emb", "conv1_wp", "conv2_wp")]


class DecoderHead(C.Structure):
    _fields_ = [("proj_wp", fp), ("proj_b", fp), ("ln_g", fp), ("ln_b", fp), ("d4", C.c_int), ("dx2", C.c_int), ("proj_w", fp)]


class DecoderWeights(C.Structure):
    _fields_ = [("proj_w", fp), ("proj_b", fp), ("proj_ln_g", fp), ("proj_ln_b", fp)] + \
               [(n, fp * MAX_DEC_LAYERS) for n in ("dw_w", "dw_b", "pw_w", "pw_b", "ln_g", "ln_b", "skip_g", "skip_b")] + \
               [("mel_w", fp), ("mel_b", fp)]


class DecoderShape(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("d4", "dx2", "kernel", "n_blocks", "block_depth", "n_mel")]


HIFIGAN_MAX_UP, HIFIGAN_MAX_KERNELS, HIFIGAN_MAX_RBCONV = 8, 8, 96


class HifiGanWeights(C.Structure):
    _fields_ = [("pre_w", fp), ("pre_b", fp), ("up_w", fp * HIFIGAN_MAX_UP), ("up_b", fp * HIFIGAN_MAX_UP),
                ("rb_w1", fp * HIFIGAN_MAX_RBCONV), ("rb_b1", fp * HIFIGAN_MAX_RBCONV), ("rb_w2", fp * HIFIGAN_MAX_RBCONV),
                ("rb_b2", fp * HIFIGAN_MAX_RBCONV), ("post_w", fp), ("post_b", fp),
                ("rb_wp1", fp * HIFIGAN_MAX_RBCONV), ("rb_wp2", fp * HIFIGAN_MAX_RBCONV)]


class HifiGanShape(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n_mel", "initial_channel", "n_up", "n_kernels", "resblock")] + \
               [("up_rates", C.c_int * HIFIGAN_MAX_UP), ("up_kernels", C.c_int * HIFIGAN_MAX_UP),
                ("rb_kernels", C.c_int * HIFIGAN_MAX_KERNELS), ("rb_dilations", C.c_int * (HIFIGAN_MAX_KERNELS * 3))]


class ConvDesc(C.Structure):
    """esmi_conv_desc (include/esmi.h): one convolution of the training step, checkpoint weight layout."""
    _fields_ = [(n, C.c_int) for n in ("B", "n_in", "c_in", "n_out", "c_out", "k", "stride", "pad", "groups", "transposed", "precision", "act")] + \
               [("packed_fwd", C.c_void_p), ("packed_grad", C.c_void_p)]


class ReduceItem(C.Structure):
    _fields_ = [("partial", fp), ("n", C.c_int64), ("stride", C.c_int64), ("chunks", C.c_int64), ("out", fp), ("n0", C.c_int64), ("out1", fp)]


class ReduceQueue(C.Structure):
    """esmi_reduce_queue (include/esmi.h): the queued second stages of a training step's chunked reductions."""
    _fields_ = [("count", C.c_int32), ("items", ReduceItem * 48)]


class TrainLossArgs(C.Structure):
    """esmi_train_loss_args (include/esmi.h): model.py:167-216."""
    _fields_ = [(n, fp) for n in ("mel_pred", "mel", "pitch_pred", "pitch", "energy_pred", "energy", "dur_pred", "dur", "mel_mask",
                                  "ph_mask")] + [(n, C.c_int) for n in ("B", "T", "L", "n_mel")] + \
               [(n, fp) for n in ("out", "d_mel", "d_pitch", "d_energy", "d_dur", "scratch", "grad_seed")]


class ForwardArgs(C.Structure):
    """esmi_forward_args (include/esmi.h): the whole inference forward behind one call."""
    _fields_ = [(n, C.c_int) for n in ("B", "T", "depth", "dim", "fuse_kernel", "plan")] + \
               [("blocks", EncoderBlockWeights * MAX_DEPTH), ("shapes", EncoderBlockShape * MAX_DEPTH), ("embed", fp),
                ("fuse", FuseWeights), ("pitch", PredictorWeights), ("energy", PredictorWeights), ("duration", PredictorWeights),
                ("head", DecoderHead), ("dec_blob", fp), ("dec_shape", DecoderShape),
                ("ids", fp), ("mask", fp), ("dur_forced", fp),
                ("duration_pred", fp), ("mel_len", fp), ("mel", fp), ("L_out", C.c_int), ("lmax_host", C.c_int), ("lmax_dev", fp),
                ("pitch_pred", fp), ("energy_pred", fp), ("pitch_idx", fp), ("energy_idx", fp), ("dur", fp), ("cum", fp),
                ("arena", fp), ("arena_bytes", C.c_size_t), ("range_flag", fp)]


# launch-plan bits (include/esmi.h ESMI_FUSE_*): passed per call, the library keeps no state
FUSE_MERGE_QKV, FUSE_ATTN_FFN, FUSE_VARIANCE, FUSE_SPLIT2, FUSE_BLOCK, FUSE_CHAIN16, FUSE_ALL = 1, 2, 4, 8, 16, 32, 63


_tls = threading.local()


def current_plan():
    """Launch plan the module mirrors pass to the C-ABI on this thread (default: everything fused)."""
    return getattr(_tls, "plan", FUSE_ALL)


@contextlib.contextmanager
def launch_plan(mask):
    """Thread-local override of the launch plan (tests / ablation): `with _lib.launch_plan(0): net(x)` runs one kernel per
    reference op.  Nothing global is touched: the mask travels as an argument of each C-ABI call."""
    old = current_plan()
    _tls.plan = int(mask) & FUSE_ALL
    try:
        yield
    finally:
        _tls.plan = old

EXPORTS = (
    "esmi_version", "esmi_backend", "esmi_build_config", "esmi_fuse_variance_adaptor_workspace_bytes",
    "esmi_fuse_variance_adaptor_f32", "esmi_decoder_head_f32", "esmi_pack_conv_weight_f32", "esmi_pack_convT_weight_f32",
    "esmi_encoder_block_workspace_bytes", "esmi_encoder_block_f32", "esmi_pool_mask_u8",
    "esmi_fuse_workspace_bytes", "esmi_fuse_f32", "esmi_variance_adaptor_workspace_bytes",
    "esmi_variance_adaptor_f32", "esmi_length_regulate_i32", "esmi_length_regulator_indices_i32",
    "esmi_upsample_f32", "esmi_mel_decoder_blob_bytes", "esmi_mel_decoder_pack_f32", "esmi_mel_decoder_f32", "esmi_mel_decoder_workspace_bytes", "esmi_mel_decoder_clock_probe",
    "esmi_mask_rows_f32", "esmi_pack_bfrag_floats", "esmi_pack_bfrag_f32", "esmi_compose_merge_f32", "esmi_max_i32",
    "esmi_self_attention_workspace_bytes", "esmi_self_attention_f32", "esmi_mixffn_workspace_bytes", "esmi_mixffn_f32",
    "esmi_acoustic_decoder_f32", "esmi_bucket_embedding_f32", "esmi_split_weight_limit", "esmi_absmax_f32",
    "esmi_forward_arena_bytes", "esmi_phoneme2mel_forward_f32", "esmi_hifigan_workspace_bytes", "esmi_hifigan_generator_f32",
    "esmi_pack_resblock_bytes", "esmi_pack_resblock_f16",
    "esmi_train_conv_fwd_f32", "esmi_train_conv_ln_fwd_f32", "esmi_train_conv_dgrad_f32", "esmi_train_conv_wgrad_f32", "esmi_train_layernorm_fwd_f32",
    "esmi_train_conv_wgrad_workspace_bytes", "esmi_train_layernorm_bwd_workspace_bytes", "esmi_train_conv_workspace_bytes",
    "esmi_train_embedding_bwd_workspace_bytes",
    "esmi_train_layernorm_bwd_f32", "esmi_train_act_fwd_f32", "esmi_train_act_bwd_f32", "esmi_train_attention_fwd_f32",
    "esmi_train_attention_bwd_f32", "esmi_train_embedding_fwd_f32", "esmi_train_embedding_bwd_f32", "esmi_train_mask_rows_f32",
    "esmi_train_add_f32", "esmi_train_copy_cols_f32", "esmi_train_repeat_fwd_f32", "esmi_train_repeat_bwd_f32",
    "esmi_train_loss_f32", "esmi_train_adamw_f32", "esmi_train_adamw_graph_f32", "esmi_train_conv_bwd_workspace_bytes",
    "esmi_train_conv_bwd_f32", "esmi_train_reduce_flush_f32", "esmi_train_pack_weights_f32", "esmi_train_cat_f32",
)


_BIND_COUNT = itertools.count(1)


def bind(lib):
    """Declare argument / return types of every entry point of include/esmi.h on `lib`."""
    lib._esmi_generation = next(_BIND_COUNT)      # identity of this binding: packed weight blobs are only valid for the build that made them
    i, sz, P = C.c_int, C.c_size_t, C.POINTER
    lib.esmi_version.restype = i
    lib.esmi_backend.restype = C.c_char_p
    lib.esmi_build_config.restype = C.c_char_p
    lib.esmi_pack_conv_weight_f32.argtypes = [fp, fp, i, i, i, fp]
    lib.esmi_pack_convT_weight_f32.argtypes = [fp, fp, i, i, i, fp]
    lib.esmi_pack_bfrag_floats.argtypes = [i, i, i]
    lib.esmi_pack_bfrag_floats.restype = sz
    lib.esmi_pack_bfrag_f32.argtypes = [fp, fp, i, i, i, fp]
    lib.esmi_compose_merge_f32.argtypes = [fp, fp, i, i, i, fp, fp]
    lib.esmi_encoder_block_workspace_bytes.argtypes = [P(EncoderBlockShape)]
    lib.esmi_encoder_block_workspace_bytes.restype = sz
    lib.esmi_encoder_block_f32.argtypes = [P(EncoderBlockWeights), P(EncoderBlockShape), fp, fp, fp, fp, fp, fp, sz, fp]
    lib.esmi_pool_mask_u8.argtypes = [fp, i, i, i, fp, i, fp]
    lib.esmi_fuse_workspace_bytes.argtypes = [i, i, i, i]
    lib.esmi_fuse_workspace_bytes.restype = sz
    lib.esmi_fuse_f32.argtypes = [P(FuseWeights), i, i, i, i, i, P(fp), P(i), fp, fp, i, fp, sz, fp]
    lib.esmi_variance_adaptor_workspace_bytes.argtypes = [i, i, i]
    lib.esmi_variance_adaptor_workspace_bytes.restype = sz
    lib.esmi_variance_adaptor_f32.argtypes = [P(PredictorWeights)] * 3 + [i, i, i] + [fp] * 11 + [fp, sz, fp]
    lib.esmi_fuse_variance_adaptor_workspace_bytes.argtypes = [i, i, i, i]
    lib.esmi_fuse_variance_adaptor_workspace_bytes.restype = sz
    lib.esmi_fuse_variance_adaptor_f32.argtypes = [P(FuseWeights), i, i, i, i, i, P(fp), P(i)] + [P(PredictorWeights)] * 3 + \
        [fp] * 13 + [P(DecoderHead), fp, i] + [fp, sz, fp]
    lib.esmi_decoder_head_f32.argtypes = [P(DecoderHead), C.c_long, fp, fp, fp]
    lib.esmi_max_i32.argtypes = [fp, i, fp, fp]
    lib.esmi_length_regulate_i32.argtypes = [fp, i, i, fp, fp, fp, fp]
    lib.esmi_length_regulator_indices_i32.argtypes = [fp, i, i, i, fp, fp]
    lib.esmi_upsample_f32.argtypes = [fp, fp, fp, i, i, i, i, fp, fp, fp]
    lib.esmi_mel_decoder_blob_bytes.argtypes = [P(DecoderShape)]
    lib.esmi_mel_decoder_blob_bytes.restype = sz
    lib.esmi_mel_decoder_pack_f32.argtypes = [P(DecoderWeights), P(DecoderShape), fp, fp]
    lib.esmi_mel_decoder_f32.argtypes = [fp, P(DecoderShape), fp, fp, fp, fp, fp, i, i, i, i, i, fp, fp, sz, fp]
    lib.esmi_mel_decoder_workspace_bytes.argtypes = [P(DecoderShape), i, i]
    lib.esmi_mel_decoder_workspace_bytes.restype = sz
    lib.esmi_mel_decoder_clock_probe.argtypes = [fp]
    lib.esmi_mask_rows_f32.argtypes = [fp, fp, C.c_int64, i, fp]
    lib.esmi_self_attention_workspace_bytes.argtypes = [i, i, i, i]
    lib.esmi_self_attention_workspace_bytes.restype = sz
    lib.esmi_self_attention_f32.argtypes = [fp, fp, fp, i, i, i, i, fp, fp, fp, sz, fp]
    lib.esmi_mixffn_workspace_bytes.argtypes = [i, i, i, i]
    lib.esmi_mixffn_workspace_bytes.restype = sz
    lib.esmi_mixffn_f32.argtypes = [fp] * 6 + [i, i, i, i, fp, fp, fp, sz, fp]
    lib.esmi_acoustic_decoder_f32.argtypes = [P(PredictorWeights), i, i, i, i, fp, i, fp, fp, fp, sz, fp]
    lib.esmi_bucket_embedding_f32.argtypes = [fp, fp, fp, C.c_int64, i, fp, fp, fp]
    lib.esmi_split_weight_limit.argtypes = []
    lib.esmi_split_weight_limit.restype = C.c_float
    lib.esmi_absmax_f32.argtypes = [fp, C.c_int64, fp, fp]
    lib.esmi_forward_arena_bytes.argtypes = [P(ForwardArgs)]
    lib.esmi_forward_arena_bytes.restype = sz
    lib.esmi_phoneme2mel_forward_f32.argtypes = [P(ForwardArgs), i, fp]
    i64, f, dbl = C.c_int64, C.c_float, C.c_double
    lib.esmi_train_conv_fwd_f32.argtypes = [P(ConvDesc), fp, fp, fp, fp, fp, sz, fp]
    lib.esmi_train_conv_ln_fwd_f32.argtypes = [P(ConvDesc), fp, fp, fp, fp, fp, fp, fp, i, fp, fp, fp, fp, fp, sz, fp]
    lib.esmi_train_conv_workspace_bytes.argtypes = [P(ConvDesc)]
    lib.esmi_train_conv_workspace_bytes.restype = sz
    lib.esmi_train_conv_dgrad_f32.argtypes = [P(ConvDesc), fp, fp, fp, fp, sz, fp]
    lib.esmi_train_conv_wgrad_f32.argtypes = [P(ConvDesc), fp, fp, fp, fp, fp, sz, fp]
    lib.esmi_train_conv_wgrad_workspace_bytes.argtypes = [P(ConvDesc)]
    lib.esmi_train_conv_wgrad_workspace_bytes.restype = sz
    lib.esmi_train_conv_bwd_workspace_bytes.argtypes = [P(ConvDesc)]
    lib.esmi_train_conv_bwd_workspace_bytes.restype = sz
    lib.esmi_train_conv_bwd_f32.argtypes = [P(ConvDesc), fp, fp, fp, fp, fp, fp, fp, sz, P(ReduceQueue), fp]
    lib.esmi_train_reduce_flush_f32.argtypes = [P(ReduceQueue), fp]
    lib.esmi_train_pack_weights_f32.argtypes = [P(ConvDesc), P(C.c_void_p), i, fp]
    lib.esmi_train_cat_f32.argtypes = [P(C.c_void_p), P(i), i, i64, fp, fp, C.c_uint, i, fp]
    lib.esmi_train_layernorm_bwd_workspace_bytes.argtypes = [i64, i]
    lib.esmi_train_layernorm_bwd_workspace_bytes.restype = sz
    lib.esmi_train_layernorm_fwd_f32.argtypes = [fp, fp, fp, i64, i, fp, fp, fp, fp, fp, fp, i, fp]
    lib.esmi_train_layernorm_bwd_f32.argtypes = [fp, fp, fp, fp, fp, i64, i, fp, fp, fp, fp, sz, P(ReduceQueue), fp, i, fp, fp]
    lib.esmi_train_act_fwd_f32.argtypes = [fp, i64, i, fp, fp]
    lib.esmi_train_act_bwd_f32.argtypes = [fp, fp, i64, i, fp, fp]
    lib.esmi_train_attention_fwd_f32.argtypes = [fp, i, i, i, i, fp, fp, fp]
    lib.esmi_train_attention_bwd_f32.argtypes = [fp, fp, fp, i, i, i, i, fp, fp, fp]
    lib.esmi_train_embedding_fwd_f32.argtypes = [fp, fp, i64, i, i, fp, fp]
    lib.esmi_train_embedding_bwd_f32.argtypes = [fp, fp, i64, i, i, i, fp, fp, sz, P(ReduceQueue), fp]
    lib.esmi_train_embedding_bwd_workspace_bytes.argtypes = [i64, i, i]
    lib.esmi_train_embedding_bwd_workspace_bytes.restype = sz
    lib.esmi_train_mask_rows_f32.argtypes = [fp, fp, i64, i, fp, fp]
    lib.esmi_train_add_f32.argtypes = [fp, fp, i64, fp, fp]
    lib.esmi_train_copy_cols_f32.argtypes = [fp, i, i, fp, i, i, i64, i, fp]
    lib.esmi_train_repeat_fwd_f32.argtypes = [fp, fp, i, i, i, i, fp, fp]
    lib.esmi_train_repeat_bwd_f32.argtypes = [fp, fp, i, i, i, i, fp, fp]
    lib.esmi_train_loss_f32.argtypes = [P(TrainLossArgs), fp]
    lib.esmi_train_adamw_f32.argtypes = [fp, fp, fp, fp, i64, dbl, dbl, dbl, dbl, dbl, i, dbl, fp]
    lib.esmi_train_adamw_graph_f32.argtypes = [fp, fp, fp, fp, i64, fp, dbl, dbl, dbl, dbl, fp, fp, fp, fp]
    lib.esmi_pack_resblock_bytes.argtypes = [i, i]
    lib.esmi_pack_resblock_bytes.restype = sz
    lib.esmi_pack_resblock_f16.argtypes = [fp, fp, i, i, fp]
    lib.esmi_hifigan_workspace_bytes.argtypes = [P(HifiGanShape), i, i]
    lib.esmi_hifigan_workspace_bytes.restype = sz
    lib.esmi_hifigan_generator_f32.argtypes = [P(HifiGanWeights), P(HifiGanShape), fp, i, i, fp, fp, sz, fp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int and name != "esmi_version":
            fn.errcheck = _make_check(name)
    return lib


class ActivationRange(ValueError):
    """ESMI_ERR_RANGE (checked build): a value entering a split-f16 contraction was outside the binary16 range."""


class Unsupported(RuntimeError):
    """ESMI_ERR_UNSUPPORTED: the shape is outside what this entry point's kernels are built for."""


def _make_check(name):
    def check(rc, func, args):
        if rc == -2:
            raise Unsupported(f"{name}: shape not supported")
        if rc == -4:
            raise ActivationRange(f"{name}: an activation is outside the binary16 range (|a| >= 65504) of the fp32-accurate split "
                                  "contractions; run this checkpoint on libesmi_fp32mfma.so (ESMI_LIB=...), which has no range limit")
        if rc != ESMI_OK:
            what = _ERRS.get(rc, f"hipError_t {rc}" if rc > 0 else f"error {rc}")
            raise RuntimeError(f"{name} failed: {what}")
        return rc
    return check


_LIB = None


def load():
    """Return the bound HIP library; raise if it is not built (no silent fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the esmi HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for this path.")
        lib = bind(C.CDLL(LIB_PATH))
        if lib.esmi_backend().decode() != "hip:gfx950":    # (tests bind the wave simulator through bind(), never through here)
            raise RuntimeError(f"{LIB_PATH} reports backend {lib.esmi_backend().decode()!r}: the product path only runs the HIP "
                               "build for gfx950 (no CPU / simulator fallback)")
        _check_toolchain(lib)
        _LIB = lib
    return _LIB


# the HIP / clang version the compiler work-arounds inside the chain kernels were validated on (esmi_build_config, esmi_abi.hip)
VALIDATED_HIP = "7.2"


def _check_toolchain(lib):
    """Warn (once, at load) when the library was built by another ROCm release than the one its two compiler work-arounds were validated
    on: the parity tests (`pytest -m gpu`) and `ESMI_DEBUG_RANGE=1` are the check to run on a new toolchain."""
    cfg = lib.esmi_build_config().decode()
    hip = next((f[4:] for f in cfg.split(",") if f.startswith("hip=")), None)
    if hip is not None and not hip.startswith(VALIDATED_HIP + "."):
        import warnings
        warnings.warn(f"libesmi.so was built with HIP {hip}; the encoder-side chain kernels carry two compiler work-arounds validated on "
                      f"ROCm {VALIDATED_HIP} (csrc/chain16.h, profiles/r05_probes/fuse_va_wrong_rows.md): run `pytest tests -m gpu` on this "
                      "toolchain before trusting the output", RuntimeWarning, stacklevel=3)


@contextlib.contextmanager
def use_library(path):
    """Bind another build of the SAME ABI (e.g. libesmi_fp32mfma.so) for the duration of the block.  Packed weight blobs are
    only valid for the build that made them (the split-f16 and the exact-fp32 build store different blobs): the modules' pack caches
    key on the binding (`generation()`), so a module used on both sides of the switch re-packs."""
    global _LIB
    old = _LIB
    new = bind(C.CDLL(os.path.abspath(path)))
    if new.esmi_backend().decode() != "hip:gfx950":
        raise RuntimeError(f"{path} reports backend {new.esmi_backend().decode()!r}: not a HIP build for gfx950")
    _LIB = new
    try:
        yield _LIB
    finally:
        _LIB = old


def generation():
    """Identity of the currently bound library (changes with `use_library`)."""
    return load()._esmi_generation


def backend(lib=None):
    return (lib or load()).esmi_backend().decode()
