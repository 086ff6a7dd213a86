"""Shared by the micro-benchmarks: trailing `name=value` arguments (persistent=0, persistent_grid=8, narrow_conv=0, narrow_mfma=0,
igemm_general_operands=1, igemm2_waves=4) become e2eft_set_option calls — the library itself never reads the environment."""
import sys

from diffusion_e2e_ft_amd import _lib

NAMES = {"persistent": _lib.OPT_PERSISTENT, "persistent_grid": _lib.OPT_PERSISTENT_GRID, "narrow_conv": _lib.OPT_NARROW_CONV,
         "narrow_mfma": _lib.OPT_NARROW_MFMA, "igemm_general_operands": _lib.OPT_IGEMM_GENERAL_OPERANDS, "igemm2_waves": _lib.OPT_IGEMM2_WAVES, "patch_conv": _lib.OPT_PATCH_CONV, "thin_input_conv": _lib.OPT_THIN_INPUT_CONV, "fused_norm": _lib.OPT_FUSED_NORM, "attn_dma": _lib.OPT_ATTN_DMA, "upconv_phases": _lib.OPT_UPCONV_PHASES, "patch_conv_2x2": _lib.OPT_PATCH_CONV_2X2, "persistent_min_qrounds": _lib.OPT_PERSISTENT_MIN_QROUNDS, "gn_apply_iters": _lib.OPT_GN_APPLY_ITERS, "f32_split": _lib.OPT_F32_SPLIT}


def take(argv):
    """strip `name=value` items from argv, apply them, return the remaining positional arguments"""
    rest = []
    for a in argv:
        k, _, v = a.partition("=")
        if k in NAMES and v.lstrip("-").isdigit():
            _lib.set_option(NAMES[k], int(v))
            print("option %s = %s" % (k, v), file=sys.stderr)      # (stderr: bench.py's stdout is ONE JSON line)
        else:
            rest.append(a)
    return rest
