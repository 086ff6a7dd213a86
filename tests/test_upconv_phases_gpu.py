"""e2eft_upconv2x_fwd (round 5): `nearest 2x upsample -> conv 3x3 / pad 1` (diffusers Upsample2D: the upsamplers of the UNet and of the VAE decoder,
unet_2d_blocks.py / the reference's `vae.decoder`) as four 2x2 convolutions of the low-resolution input, one per output parity, written interleaved into the
full-resolution tensor by the persistent kernel (IgemmParams.out_seg).  Checked against torch (fp64 upsample + conv), against the fused-upsample 3x3 form it
replaces, and through the GroupNorm that consumes the statistics its epilogue emits.  Shapes: segment width 16 ... 384, 64-row blocks that start inside a
segment and cross one or several segment ends, one and two N tiles, bias, fp16 / bf16.  E2EFT_OPT_PERSISTENT_GRID = 8 sends the small cases through the kernel."""
import pytest
import torch
import torch.nn.functional as TF

from util import assert_close, nhwc, pack_conv_weight, q, rel_err, to_nchw

pytestmark = pytest.mark.gpu
GRID_FROM_ENV = "E2EFT_TEST_PERSISTENT_GRID" in __import__("os").environ      # tests/conftest.py: the whole-suite variant that sends small problems through the persistent kernels


def _launches():
    import ctypes
    from diffusion_e2e_ft_amd import _lib
    lib = _lib.load()
    lib.e2eft_debug_persistent_launches.restype = ctypes.c_long
    lib.e2eft_debug_patch_launches.restype = ctypes.c_long
    return lib.e2eft_debug_persistent_launches() + lib.e2eft_debug_patch_launches()      # igemm5, or (round 6, eligible shapes) igemm6's 2x2-tap variant


def _case(dev, dtype, B, H, W, Ci, Co, seed, bias=True):
    from diffusion_e2e_ft_amd import autograd as F
    g = torch.Generator().manual_seed(seed)
    conv = torch.nn.Conv2d(Ci, Co, 3, padding=1, bias=bias)
    with torch.no_grad():
        conv.weight.copy_(q(torch.randn(conv.weight.shape, generator=g) / (9 * Ci) ** 0.5, dtype))
        if bias:
            conv.bias.copy_(q(torch.randn(Co, generator=g), dtype))
    x = q(torch.randn(B, Ci, H, W, generator=g), dtype)
    ref = conv.double()(TF.interpolate(x.double(), scale_factor=2.0, mode="nearest")).float()
    conv = conv.float().to(dev)
    xd = nhwc(x, dtype, dev)
    wd = pack_conv_weight(conv.weight.detach().cpu(), dtype, dev)
    bd = conv.bias.detach().to(dtype) if bias else None
    return conv, xd, wd, bd, ref, (lambda: F.phase_conv_weight(conv, dtype))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 32, 32, 64, 256), (3, 16, 48, 128, 256), (6, 16, 16, 64, 320), (2, 16, 80, 64, 256), (1, 64, 64, 64, 128), (8, 16, 16, 192, 256)])
def test_phases_match_torch_and_the_fused_upsample_form(dev, dtype, B, H, W, Ci, Co):
    from diffusion_e2e_ft_amd import ops, _lib
    conv, xd, wd, bd, ref, wph = _case(dev, dtype, B, H, W, Ci, Co, seed=H * 7 + W + Ci)
    with _lib.option(_lib.OPT_PERSISTENT_GRID, 8):
        before = _launches()
        y = ops.conv2d(xd, wd, bd, Co, 3, 3, 1, (1, 1, 1, 1), up_to=(2 * H, 2 * W), gn_stats=True, w_phase=wph)
        ran = _launches() - before
        with _lib.option(_lib.OPT_UPCONV_PHASES, 0):
            y0 = ops.conv2d(xd, wd, bd, Co, 3, 3, 1, (1, 1, 1, 1), up_to=(2 * H, 2 * W), gn_stats=True, w_phase=wph)
    torch.cuda.synchronize()
    assert ran == 4, "the four phase launches did not run on the persistent kernel (%d)" % ran
    assert tuple(y.shape) == (B, 2 * H, 2 * W, Co)
    assert_close(to_nchw(y), ref, dtype, "upconv2x phases vs torch")
    assert_close(to_nchw(y0), ref, dtype, "fused-upsample form vs torch")
    # the two forms differ by the rounding of the summed weights and the summation order only: well inside the single-op bar of the dtype
    assert rel_err(to_nchw(y), to_nchw(y0)) <= {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    # statistics emitted by the four phases -> the consuming GroupNorm (+ SiLU) equals the one that recomputes them from the tensor, and torch
    st = getattr(y, "_e2eft_gn", None)
    assert st is not None and st.nslabs == 4 * H * W // 256
    g = torch.Generator().manual_seed(1)
    ga, be = q(1 + 0.3 * torch.randn(Co, generator=g), dtype), q(0.3 * torch.randn(Co, generator=g), dtype)
    gn = ops.groupnorm(y, ga.to(dtype).to(dev), be.to(dtype).to(dev), 32, 1e-6, True)
    ops.GN_STATS_ENABLED = False
    try:
        gn2 = ops.groupnorm(y, ga.to(dtype).to(dev), be.to(dtype).to(dev), 32, 1e-6, True)
    finally:
        ops.GN_STATS_ENABLED = True
    assert_close(to_nchw(gn), to_nchw(gn2), dtype, "phase statistics vs stand-alone statistics", scale=0.5)
    want = TF.silu(TF.group_norm(to_nchw(y).double(), 32, ga.double(), be.double(), 1e-6)).float()
    assert_close(to_nchw(gn), want, dtype, "groupnorm on phase statistics", scale=1.5)


def _labels(fn):
    from diffusion_e2e_ft_amd import ops
    timer = ops.KernelTimer()
    ops.TIMER = timer
    try:
        y = fn()
        torch.cuda.synchronize()
    finally:
        ops.TIMER = None
    return y, [lab[0] if isinstance(lab, tuple) else lab for (_, lab) in timer.by_label()]


@pytest.mark.parametrize("dtype,B,H,W,Ci,Co", [(torch.float32, 4, 32, 32, 32, 64), (torch.float32, 6, 24, 40, 64, 128), (torch.float16, 8, 24, 24, 64, 128),
                                                (torch.bfloat16, 16, 9, 30, 128, 128)])
def test_phases_on_the_general_kernel_fp32_and_odd_widths(dev, dtype, B, H, W, Ci, Co):
    """shapes the persistent kernel does not take (fp32; widths that are not multiples of 16) run the four phases on igemm2, whose row passes place every row by
    division — no statistics there (the consuming GroupNorm runs its own pass).  (Like the persistent route it wants two tiles per CU: the test grid of 8 CUs.)"""
    from diffusion_e2e_ft_amd import ops, _lib
    from util import TOL
    conv, xd, wd, bd, ref, wph = _case(dev, dtype, B, H, W, Ci, Co, seed=H + W)
    with _lib.option(_lib.OPT_PERSISTENT_GRID, 8):
        y, labels = _labels(lambda: ops.conv2d(xd, wd, bd, Co, 3, 3, 1, (1, 1, 1, 1), up_to=(2 * H, 2 * W), gn_stats=True, w_phase=wph))
        y0, labels0 = _labels(lambda: ops.conv2d(xd, wd, bd, Co, 3, 3, 1, (1, 1, 1, 1), up_to=(2 * H, 2 * W), gn_stats=True))      # no phase weights offered: the fused-upsample form
    assert any(str(l).startswith("upconv2x") for l in labels), labels
    assert getattr(y, "_e2eft_gn", None) is None
    assert_close(to_nchw(y), ref, dtype, "phases on igemm2")
    assert not any(str(l).startswith("upconv2x") for l in labels0)
    assert rel_err(to_nchw(y), to_nchw(y0)) <= 2 * TOL[dtype]
    if GRID_FROM_ENV:
        return          # (a process-wide test grid — the suite's E2EFT_TEST_PERSISTENT_GRID variant — leaves no "real machine" to fall back on)
    # the same shape without the test grid: fewer than two tiles per CU of the real machine -> the fused-upsample form, silently
    y1, labels1 = _labels(lambda: ops.conv2d(xd, wd, bd, Co, 3, 3, 1, (1, 1, 1, 1), up_to=(2 * H, 2 * W), w_phase=wph))
    assert not any(str(l).startswith("upconv2x") for l in labels1) and rel_err(to_nchw(y1), to_nchw(y0)) <= TOL[dtype]


def test_unsupported_shapes_fall_back_and_the_entry_point_says_so(dev):
    from diffusion_e2e_ft_amd import ops, _lib
    dtype = torch.float16
    # 8 input channels (less than one k-tile of the FAST operand path): the fused-upsample form serves it, silently
    conv, xd, wd, bd, ref, wph = _case(dev, dtype, 1, 32, 32, 8, 128, seed=3)
    y, labels = _labels(lambda: ops.conv2d(ops.pad_channels(xd), wd, bd, 128, 3, 3, 1, (1, 1, 1, 1), up_to=(64, 64), w_phase=wph))
    assert_close(to_nchw(y), ref, dtype, "fallback")
    assert not any(str(l).startswith("upconv2x") for l in labels), labels
    d = ops._conv_desc(xd, None, 128, 3, 3, 1, (1, 1, 1, 1), (64, 64), 1.0, ldo=128)
    import ctypes as C
    lib = _lib.load()
    assert lib.e2eft_upconv2x_fwd_supported(C.byref(d)) == 0
    out = torch.empty((1, 64, 64, 128), dtype=dtype, device=dev)
    rc = lib.e2eft_upconv2x_fwd(C.byref(d), xd.data_ptr(), xd.data_ptr(), None, out.data_ptr(), None, 0, None, None)
    assert rc == 4 and b"not eligible" in lib.e2eft_last_error()
    conv, xd, wd, bd, ref, wph = _case(dev, dtype, 2, 32, 32, 64, 256, seed=5)
    d2 = ops._conv_desc(xd, None, 256, 3, 3, 1, (1, 1, 1, 1), (64, 64), 1.0, ldo=256)
    if not GRID_FROM_ENV:
        assert lib.e2eft_upconv2x_fwd_supported(C.byref(d2)) == 0      # 16 tiles per phase: fewer than two per CU of the real machine
    with _lib.option(_lib.OPT_PERSISTENT_GRID, 8):
        assert lib.e2eft_upconv2x_fwd_supported(C.byref(d2)) == 1      # ... enough for the 8-CU test grid
        with _lib.option(_lib.OPT_UPCONV_PHASES, 0):                    # the switch: supported() answers 0 for an eligible shape
            assert lib.e2eft_upconv2x_fwd_supported(C.byref(d2)) == 0
        assert lib.e2eft_upconv2x_fwd_supported(C.byref(d2)) == 1


def test_vae_decoder_upsampler_at_full_width_runs_in_phases_and_matches(dev):
    """the layer the route was built for, at a width of the benchmarked configuration: 256 -> 256 channels from 192 x 384 to 384 x 768 (one image: the reference
    convolution is evaluated on the host), with its bias; phases must have run WITHOUT the test grid"""
    from diffusion_e2e_ft_amd import ops, _lib
    dtype = torch.float16
    conv, xd, wd, bd, ref, wph = _case(dev, dtype, 1, 192, 384, 256, 256, seed=9)
    before = _launches()
    y = ops.conv2d(xd, wd, bd, 256, 3, 3, 1, (1, 1, 1, 1), up_to=(384, 768), gn_stats=True, w_phase=wph)
    torch.cuda.synchronize()
    assert _launches() - before == 4
    assert_close(to_nchw(y), ref, dtype, "256 -> 256 upsampler, 192x384 -> 384x768")
    y2 = ops.conv2d(xd, wd, bd, 256, 3, 3, 1, (1, 1, 1, 1), up_to=(384, 768), gn_stats=True, w_phase=wph)
    assert torch.equal(y, y2) and torch.equal(y._e2eft_gn.partial, y2._e2eft_gn.partial)          # deterministic run to run
