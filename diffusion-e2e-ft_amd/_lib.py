"""ctypes binding of libe2eft.so (include/e2eft.h).  Fails loudly when the library is missing: there is no
PyTorch / CPU fallback for the compute path."""
import ctypes as C
import os

import torch  # noqa: F401  (must be imported first so that libamdhip64.so.7 is torch's copy)

from . import build as _build

_LIB = None

F32, F16, BF16 = 0, 1, 2
_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}


def dtype_id(dt):
    try:
        return _DT[dt]
    except KeyError:
        raise TypeError("libe2eft supports float32/float16/bfloat16, got %s" % dt)


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dtype", "batch", "hin", "win", "hl", "wl", "c1", "ldx1", "c2", "ldx2", "kh", "kw", "stride", "pad_t", "pad_l",
        "hout", "wout", "cout", "ldo", "ldr", "ldw")] + [("alpha", C.c_float)]


class GemmDesc(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("dtype", "m", "n", "k", "lda", "ldw", "ldo", "ldr", "nzo", "nzi")]
                + [(n, C.c_int64) for n in ("sa_o", "sa_i", "sw_o", "sw_i", "so_o", "so_i", "sr_o", "sr_i")]
                + [("bias_along_m", C.c_int32), ("alpha", C.c_float)])


class GroupNormDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dtype", "batch", "hw", "c1", "ldx1", "c2", "ldx2", "groups", "ldy", "silu")] + [
        ("eps", C.c_float)]


class AttnDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dtype", "batch", "heads", "nq", "nk_seg", "kv_nseg", "kv_bmod", "ldq", "ldk", "ldv",
                                          "ldo")] + [("scale", C.c_float)]


# e2eft_set_option keys (include/e2eft.h)
OPT_PERSISTENT, OPT_PERSISTENT_GRID, OPT_NARROW_CONV, OPT_NARROW_MFMA, OPT_IGEMM_GENERAL_OPERANDS, OPT_IGEMM2_WAVES, OPT_PATCH_CONV, OPT_THIN_INPUT_CONV, OPT_FUSED_NORM, OPT_ATTN_DMA, OPT_UPCONV_PHASES, OPT_PATCH_CONV_2X2, OPT_PERSISTENT_MIN_QROUNDS, OPT_GN_APPLY_ITERS, OPT_F32_SPLIT = range(15)

_P = C.c_void_p
_I = C.c_int32
_L = C.c_int64
_F = C.c_float
_Z = C.c_size_t

# name -> (restype, argtypes); must list every symbol declared in include/e2eft.h (tests/test_abi.py checks)
SIGNATURES = {
    "e2eft_version": (_I, []),
    "e2eft_last_error": (C.c_char_p, []),
    "e2eft_build_id": (C.c_char_p, []),
    "e2eft_set_option": (_I, [_I, _I]),
    "e2eft_get_option": (_I, [_I]),
    "e2eft_conv2d_fwd": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "e2eft_gemm": (_I, [C.POINTER(GemmDesc), _P, _P, _P, _P, _P, _P]),
    "e2eft_conv2d_fwd_gnstats": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _Z, C.POINTER(C.c_int32), _P]),
    "e2eft_conv2d_splitk_workspace_bytes": (_Z, [C.POINTER(ConvDesc)]),
    "e2eft_conv2d_fwd_splitk": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "e2eft_gemm_gnstats": (_I, [C.POINTER(GemmDesc), _P, _P, _P, _P, _P, _I, _P, _Z, C.POINTER(C.c_int32), _P]),
    "e2eft_groupnorm_fwd_pre": (_I, [C.POINTER(GroupNormDesc), _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _Z, _P]),
    "e2eft_groupnorm_coeff_offset": (_Z, [C.POINTER(GroupNormDesc)]),
    "e2eft_groupnorm_fwd_stats": (_I, [C.POINTER(GroupNormDesc), _P, _P, _P, _P, _I, _P, _I, _P, _Z, _P]),
    "e2eft_conv2d_fwd_normed_supported": (_I, [C.POINTER(ConvDesc)]),
    "e2eft_upconv2x_fwd_supported": (_I, [C.POINTER(ConvDesc)]),
    "e2eft_upconv2x_fwd": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _Z, C.POINTER(C.c_int32), _P]),
    "e2eft_groupnorm_fwd_split": (_I, [C.POINTER(GroupNormDesc), _P, _P, _P, _P, _I, _P, _P, _I, _P, _Z, _P]),
    "e2eft_f32_split_weight": (_I, [_P, _L, _I, _P, _P, _P]),
    "e2eft_f32_split2_cat": (_I, [_P, _I, _I, _P, _I, _I, _L, _P, _I, _P, _P]),
    "e2eft_f32_split2": (_I, [_P, _L, _I, _I, _P, _I, _P, _P]),
    "e2eft_conv2d_fwd_f32split_supported": (_I, [C.POINTER(ConvDesc)]),
    "e2eft_conv2d_fwd_f32split": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _Z, C.POINTER(C.c_int32), _P]),
    "e2eft_upconv2x_fwd_f32split_supported": (_I, [C.POINTER(ConvDesc)]),
    "e2eft_upconv2x_fwd_f32split": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _Z, C.POINTER(C.c_int32), _P]),
    "e2eft_gemm_f32split_supported": (_I, [C.POINTER(GemmDesc)]),
    "e2eft_gemm_f32split": (_I, [C.POINTER(GemmDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "e2eft_conv2d_fwd_normed": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _Z, C.POINTER(C.c_int32), _P]),
    "e2eft_groupnorm_workspace_bytes": (_Z, [C.POINTER(GroupNormDesc)]),
    "e2eft_groupnorm_fwd": (_I, [C.POINTER(GroupNormDesc), _P, _P, _P, _P, _P, _P, _Z, _P]),
    "e2eft_layernorm_fwd": (_I, [_I, _L, _I, _I, _I, _F, _P, _P, _P, _P, _P]),
    "e2eft_geglu_fwd": (_I, [_I, _L, _I, _I, _I, _P, _P, _P]),
    "e2eft_softmax_rows": (_I, [_I, _L, _I, _L, _F, _P, _P]),
    "e2eft_softmax_rows_causal": (_I, [_I, _L, _I, _L, _F, _I, _P, _P]),
    "e2eft_attn_fwd": (_I, [C.POINTER(AttnDesc), _P, _P, _P, _P, _P]),
    "e2eft_attn_fwd_lse": (_I, [C.POINTER(AttnDesc), _P, _P, _P, _P, _P, _P]),
    "e2eft_attn512_workspace_bytes": (_Z, [C.POINTER(AttnDesc)]),
    "e2eft_attn512_fwd": (_I, [C.POINTER(AttnDesc), _P, _P, _P, _P, _P, _Z, _P]),
    "e2eft_attn_bwd_workspace_bytes": (_Z, [C.POINTER(AttnDesc)]),
    "e2eft_attn_bwd": (_I, [C.POINTER(AttnDesc), _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, _Z, _P]),
    "e2eft_nchw_to_nhwc": (_I, [_I, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P]),
    "e2eft_nhwc_to_nchw": (_I, [_I, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P]),
    "e2eft_copy_scale": (_I, [_I, _L, _I, _I, _I, _F, _F, _P, _P, _P]),
    "e2eft_add": (_I, [_I, _L, _I, _I, _I, _I, _P, _P, _P, _P]),
    "e2eft_timestep_embedding": (_I, [_I, _I, _I, _P, _P, _P]),
    "e2eft_silu": (_I, [_I, _L, _P, _P, _P]),
    "e2eft_depth_head": (_I, [_I, _I, _L, _I, _I, _P, _P, _P]),
    "e2eft_normal_head": (_I, [_I, _I, _I, _I, _I, _I, _F, _P, _P, _P]),
    "e2eft_ssi_loss_workspace_bytes": (_Z, [_I]),
    "e2eft_ssi_loss_fwd": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "e2eft_angular_loss_workspace_bytes": (_Z, [_I]),
    "e2eft_angular_loss_fwd": (_I, [_I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    # backward / optimizer
    "e2eft_conv2d_dgrad": (_I, [C.POINTER(ConvDesc), _P, _I, _I, _P, _I, _P, _I, _P]),
    "e2eft_conv2d_wgrad_workspace_bytes": (_Z, [C.POINTER(ConvDesc), _I]),
    "e2eft_conv2d_wgrad": (_I, [C.POINTER(ConvDesc), _P, _I, _P, _P, _P, _Z, C.POINTER(C.c_int32), _P]),
    "e2eft_conv2d_im2col_t": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _L, _P]),
    "e2eft_transpose": (_I, [_I, _I, _L, _I, _L, _L, _L, _L, _L, _P, _P, _P]),
    "e2eft_colsum_workspace_bytes": (_Z, [_I, _L, _I]),
    "e2eft_colsum": (_I, [_I, _I, _L, _I, _L, _F, _P, _P, _P, _Z, _P]),
    "e2eft_upsample_nearest_bwd": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "e2eft_groupnorm_bwd_workspace_bytes": (_Z, [C.POINTER(GroupNormDesc)]),
    "e2eft_groupnorm_bwd": (_I, [C.POINTER(GroupNormDesc), _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _Z, _P]),
    "e2eft_groupnorm_bwd_add": (_I, [C.POINTER(GroupNormDesc), _P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _Z, _P]),
    "e2eft_layernorm_bwd_workspace_bytes": (_Z, [_L, _I]),
    "e2eft_layernorm_bwd": (_I, [_I, _L, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "e2eft_geglu_bwd": (_I, [_I, _L, _I, _I, _I, _I, _P, _P, _P, _P]),
    "e2eft_softmax_bwd_rows": (_I, [_I, _L, _I, _L, _F, _P, _P, _P]),
    "e2eft_silu_bwd": (_I, [_I, _L, _P, _P, _P, _P]),
    "e2eft_depth_head_bwd": (_I, [_I, _I, _L, _I, _I, _I, _I, _P, _P, _P, _P]),
    "e2eft_normal_head_bwd": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P]),
    "e2eft_ssi_loss_bwd": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "e2eft_angular_loss_bwd": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "e2eft_sumsq": (_I, [_L, _P, _P, _P]),
    "e2eft_adamw_step": (_I, [_L, _P, _P, _P, _P, _F, _F, _F, _F, _F, _I, _P, _F, _F, _P]),
    "e2eft_adamw_step_guarded": (_I, [_L, _P, _P, _P, _P, _F, _F, _F, _F, _F, _P, _P, _P, _F, _F, _P]),
    "e2eft_cast": (_I, [_I, _I, _L, _F, _I, _P, _P, _P]),
    "e2eft_ema_step": (_I, [_L, _P, _P, _F, _P]),
    "e2eft_activation": (_I, [_I, _I, _L, _P, _P, _P]),
    "e2eft_masked_quantiles_workspace_bytes": (_Z, [_I]),
    "e2eft_masked_quantiles": (_I, [_I, _L, _P, _F, _F, _F, _F, _P, _P, _Z, _P]),
    "e2eft_prepare_sample": (_I, [_I, _L, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P]),
    "e2eft_align_normals_u8": (_I, [_I, _I, _I, _P, _P, C.POINTER(C.c_double), _P, _P]),
    "e2eft_resample_bilinear_aa": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _I, _I, _F, _F, _P, _P, _P]),
    "e2eft_minmax_unit_workspace_bytes": (_Z, []),
    "e2eft_minmax_unit": (_I, [_L, _P, _P, _P, _P, _Z, _P]),
    "e2eft_aug_resample_bilinear_u8": (_I, [_I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _P, _P]),
    "e2eft_aug_gather_f32": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "e2eft_aug_gather_u8": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "e2eft_depth_eval_workspace_bytes": (_Z, [_I]),
    "e2eft_depth_eval": (_I, [_I, _I, _I, _P, _P, _P, _I, _I, _F, _F, _P, _P, _P, _Z, _P]),
    "e2eft_ensemble_workspace_bytes": (_Z, [_I]),
    "e2eft_ensemble_minmax": (_I, [_I, _L, _P, _P, _P, _Z, _P]),
    "e2eft_ensemble_gram": (_I, [_I, _L, _P, _P, _P, _P, _Z, _P]),
    "e2eft_ensemble_depth_reduce": (_I, [_I, _L, _P, _P, _P, _I, _P, _P, _P, _P, _Z, _P]),
    "e2eft_ensemble_depth_finish": (_I, [_L, _P, _P, _P, _P]),
    "e2eft_ensemble_normals": (_I, [_I, _L, _P, _P, _P, _P, _Z, _P]),
}


def lib_path():
    return os.environ.get("E2EFT_LIB", _build.LIB)   # E2EFT_LIB: load an instrumented build of the same sources (debugging only)


def load():
    """Load (building if the sources are newer and hipcc is available) and type the library."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    default = os.path.abspath(path) == os.path.abspath(_build.LIB)
    if default and os.path.exists(path) and os.path.isdir(_build.CSRC) and _build.built_id() != _build.source_id():
        # the binary on disk was built from other sources than the ones next to it (an edit without a rebuild, a stale snapshot): never run it silently
        try:
            _build.build(verbose=False)
        except Exception as e:
            raise RuntimeError("libe2eft.so (build id %s) does not match the sources in %s (id %s) and could not be rebuilt: %s"
                               % (_build.built_id(), _build.CSRC, _build.source_id(), e))
    if not os.path.exists(path):
        try:
            _build.build(verbose=False)
        except Exception as e:  # no hipcc on this box and no prebuilt library
            raise RuntimeError(
                "libe2eft.so is missing (%s) and could not be built: %s. The HIP extension is mandatory; "
                "run `python -c 'import __graft_entry__ as g; g.build()'` on a box with ROCm." % (path, e))
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.e2eft_version() < 119:
        raise RuntimeError("libe2eft.so is stale (version %d)" % lib.e2eft_version())
    if default and os.path.isdir(_build.CSRC) and lib.e2eft_build_id().decode() != _build.source_id():
        raise RuntimeError("libe2eft.so carries build id %s, the sources next to it hash to %s: rebuild (python -m diffusion_e2e_ft_amd.build --force)"
                           % (lib.e2eft_build_id().decode(), _build.source_id()))
    _LIB = lib
    return lib


def build_id():
    """e2eft_build_id() of the loaded library (hash of its sources + flags)"""
    return load().e2eft_build_id().decode()


def set_option(key, value):
    check(load().e2eft_set_option(key, value))


class option:
    """`with _lib.option(_lib.OPT_PERSISTENT, 0): ...` — scoped e2eft_set_option (A/B runs, tests)"""

    def __init__(self, key, value):
        self.key, self.value = key, value

    def __enter__(self):
        self.old = load().e2eft_get_option(self.key)
        set_option(self.key, self.value)
        return self

    def __exit__(self, *a):
        set_option(self.key, self.old)
        return False


def check(rc):
    if rc != 0:
        msg = load().e2eft_last_error()
        raise RuntimeError("libe2eft error %d: %s" % (rc, msg.decode() if msg else "?"))
