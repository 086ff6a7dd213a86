"""igemm6's 2x2-tap variant (round 6, `igemm6_kernel<T, false, false, 2>`): the halo-patch kernel of tests/test_patch_conv_gpu.py with a (8+1) x (32+1)-pixel
patch per 64-channel chunk and four k-tiles per chunk — the kernel the four parity phases of an upsampler convolution (e2eft_upconv2x_fwd: diffusers Upsample2D,
`F.interpolate(scale_factor=2, mode="nearest")` + conv3x3 in the VAE decoder's UpDecoderBlock2D and the UNet's up blocks) run on when the low-resolution width is a
multiple of 32, the height of 8 and the input has >= 128 channels in multiples of 64.
  (i)   a plain 2x2 / stride-1 convolution with each of the four (top, left) pad combinations of the phases against torch in float64 — the padding IS the
        kernel's out-of-range patch rows — incl. impulse responses across tile borders, bit-exact;
  (ii)  the upsampler through its phases on this kernel against torch (fp64 upsample + conv), against the same phases on igemm5 and against the fused-upsample
        3x3 form; GroupNorm statistics deposited by the four phases through the consuming GroupNorm;
  (iii) 2 / 3 / 5 chunks, one / two / ragged N tiles, one tile row per image, several images per XCD range, fp16 and bf16.
E2EFT_OPT_PERSISTENT_GRID = 8 sends the small cases through the persistent kernels."""
import ctypes

import pytest
import torch
import torch.nn.functional as TF

from util import assert_close, nhwc, pack_conv_weight, q, rel_err, to_nchw

pytestmark = pytest.mark.gpu


def _patch_launches():
    from diffusion_e2e_ft_amd import _lib
    lib = _lib.load()
    lib.e2eft_debug_patch_launches.restype = ctypes.c_long
    return lib.e2eft_debug_patch_launches()


def _last_kernel():
    from diffusion_e2e_ft_amd import _lib
    lib = _lib.load()
    lib.e2eft_debug_last_kernel.restype = ctypes.c_char_p
    return lib.e2eft_debug_last_kernel().decode()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("pt,pl", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("B,H,W,Ci,Co", [(4, 32, 32, 128, 128), (8, 8, 64, 192, 256), (5, 16, 32, 320, 200)])
def test_plain_2x2_convolution_every_phase_padding(dev, dtype, pt, pl, B, H, W, Ci, Co):
    """out[y, x] = sum_{i, j < 2} w[i, j] . x[y - pt + i, x - pl + j] with zeros outside the image: same-size output, pads (pt, 1 - pt, pl, 1 - pl)"""
    from diffusion_e2e_ft_amd import ops, _lib
    g = torch.Generator().manual_seed(B * 100 + H + W + Ci + 2 * pt + pl)
    x = q(torch.randn(B, Ci, H, W, generator=g), dtype)
    w = q(torch.randn(Co, Ci, 2, 2, generator=g) / (4 * Ci) ** 0.5, dtype)
    b = q(torch.randn(Co, generator=g), dtype)
    ref = TF.conv2d(TF.pad(x.double(), (pl, 1 - pl, pt, 1 - pt)), w.double(), b.double()).float()
    xd, wd, bd = nhwc(x, dtype, dev), pack_conv_weight(w, dtype, dev), b.to(dtype).to(dev)
    with _lib.option(_lib.OPT_PERSISTENT_GRID, 8):
        before = _patch_launches()
        y = ops.conv2d(xd, wd, bd, Co, 2, 2, 1, (pt, 1 - pt, pl, 1 - pl), gn_stats=True)
        assert _patch_launches() - before == 1 and _last_kernel().endswith(", false, false, 2>"), _last_kernel()
        with _lib.option(_lib.OPT_PATCH_CONV_2X2, 0):
            y5 = ops.conv2d(xd, wd, bd, Co, 2, 2, 1, (pt, 1 - pt, pl, 1 - pl), gn_stats=True)
            assert "igemm6" not in _last_kernel()
    torch.cuda.synchronize()
    assert_close(to_nchw(y), ref, dtype, "2x2 conv on igemm6 vs torch")
    assert_close(to_nchw(y5), ref, dtype, "2x2 conv on igemm5 vs torch")
    # statistics of the tile-wise deposits through the consuming GroupNorm
    ga, be = q(1 + 0.3 * torch.randn(Co, generator=g), dtype), q(0.3 * torch.randn(Co, generator=g), dtype)
    if Co % 32 == 0:
        gn = ops.groupnorm(y, ga.to(dtype).to(dev), be.to(dtype).to(dev), 32, 1e-6, True)
        want = TF.silu(TF.group_norm(to_nchw(y).double(), 32, ga.double(), be.double(), 1e-6)).float()
        assert_close(to_nchw(gn), want, dtype, "groupnorm on the 2x2 kernel's statistics", scale=1.5)


@pytest.mark.parametrize("pt,pl", [(1, 1), (0, 0), (1, 0)])
def test_impulse_responses_across_tile_borders_are_exact(dev, pt, pl):
    """one-hot inputs at tile corners / image corners and power-of-two weights: every product and sum is exact in fp16, so the output must EQUAL torch's"""
    from diffusion_e2e_ft_amd import ops, _lib
    dtype = torch.float16
    B, H, W, Ci, Co = 4, 16, 64, 128, 128
    x = torch.zeros(B, Ci, H, W)
    for (b_, c_, y_, x_) in [(0, 0, 0, 0), (0, 5, 7, 31), (0, 64, 8, 32), (1, 127, 15, 63), (3, 65, 7, 32), (2, 3, 8, 0), (3, 9, 0, 63)]:
        x[b_, c_, y_, x_] = 1.0
    g = torch.Generator().manual_seed(7)
    w = torch.pow(2.0, torch.randint(-3, 3, (Co, Ci, 2, 2), generator=g).float()) * (torch.randint(0, 2, (Co, Ci, 2, 2), generator=g).float() * 2 - 1)
    ref = TF.conv2d(TF.pad(x, (pl, 1 - pl, pt, 1 - pt)), w)
    with _lib.option(_lib.OPT_PERSISTENT_GRID, 8):
        y = ops.conv2d(nhwc(x, dtype, dev), pack_conv_weight(w, dtype, dev), None, Co, 2, 2, 1, (pt, 1 - pt, pl, 1 - pl))
        assert _last_kernel().endswith(", false, false, 2>"), _last_kernel()
    torch.cuda.synchronize()
    assert torch.equal(to_nchw(y).float().cpu(), ref)


def _upcase(dtype, B, H, W, Ci, Co, seed, dev):
    from diffusion_e2e_ft_amd import autograd as F
    g = torch.Generator().manual_seed(seed)
    conv = torch.nn.Conv2d(Ci, Co, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(q(torch.randn(conv.weight.shape, generator=g) / (9 * Ci) ** 0.5, dtype))
        conv.bias.copy_(q(torch.randn(Co, generator=g), dtype))
    x = q(torch.randn(B, Ci, H, W, generator=g), dtype)
    ref = conv.double()(TF.interpolate(x.double(), scale_factor=2.0, mode="nearest")).float()
    conv = conv.float().to(dev)
    return conv, nhwc(x, dtype, dev), pack_conv_weight(conv.weight.detach().cpu(), dtype, dev), conv.bias.detach().to(dtype), ref, (lambda: F.phase_conv_weight(conv, dtype))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 32, 32, 128, 256), (4, 16, 64, 192, 128), (6, 8, 32, 128, 320), (8, 16, 32, 320, 64), (1, 24, 96, 256, 256)])
def test_upsampler_phases_on_the_2x2_patch_kernel(dev, dtype, B, H, W, Ci, Co):
    from diffusion_e2e_ft_amd import ops, _lib
    conv, xd, wd, bd, ref, wph = _upcase(dtype, B, H, W, Ci, Co, H * 7 + W + Ci, dev)
    with _lib.option(_lib.OPT_PERSISTENT_GRID, 8):
        before = _patch_launches()
        y = ops.conv2d(xd, wd, bd, Co, 3, 3, 1, (1, 1, 1, 1), up_to=(2 * H, 2 * W), gn_stats=True, w_phase=wph)
        ran = _patch_launches() - before
        with _lib.option(_lib.OPT_PATCH_CONV_2X2, 0):
            y5 = ops.conv2d(xd, wd, bd, Co, 3, 3, 1, (1, 1, 1, 1), up_to=(2 * H, 2 * W), gn_stats=True, w_phase=wph)
        with _lib.option(_lib.OPT_UPCONV_PHASES, 0):
            y0 = ops.conv2d(xd, wd, bd, Co, 3, 3, 1, (1, 1, 1, 1), up_to=(2 * H, 2 * W), gn_stats=True, w_phase=wph)
    torch.cuda.synchronize()
    assert ran == 4, "the four phase launches did not run on the 2x2 patch kernel (%d)" % ran
    assert_close(to_nchw(y), ref, dtype, "phases on igemm6 (2x2) vs torch")
    assert_close(to_nchw(y5), ref, dtype, "phases on igemm5 vs torch")
    # same weights, same products: the two kernels differ in summation order only (chunk-major against tap-major)
    assert rel_err(to_nchw(y), to_nchw(y5)) <= {torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]
    assert rel_err(to_nchw(y), to_nchw(y0)) <= {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    st = getattr(y, "_e2eft_gn", None)
    assert st is not None and st.nslabs == 4 * H * W // 256
    if Co % 32 == 0:
        g = torch.Generator().manual_seed(1)
        ga, be = q(1 + 0.3 * torch.randn(Co, generator=g), dtype), q(0.3 * torch.randn(Co, generator=g), dtype)
        gn = ops.groupnorm(y, ga.to(dtype).to(dev), be.to(dtype).to(dev), 32, 1e-6, True)
        ops.GN_STATS_ENABLED = False
        try:
            gn2 = ops.groupnorm(y, ga.to(dtype).to(dev), be.to(dtype).to(dev), 32, 1e-6, True)
        finally:
            ops.GN_STATS_ENABLED = True
        assert_close(to_nchw(gn), to_nchw(gn2), dtype, "phase statistics vs stand-alone statistics", scale=0.5)


def test_full_width_decoder_upsampler_takes_the_2x2_patch_kernel_and_repeats_bit_exactly(dev):
    """256 -> 256 from 192 x 384 to 384 x 768 WITHOUT the test grid (the real machine): four patch launches, deterministic run to run"""
    from diffusion_e2e_ft_amd import ops
    dtype = torch.float16
    conv, xd, wd, bd, ref, wph = _upcase(dtype, 1, 192, 384, 256, 256, 9, dev)
    before = _patch_launches()
    y = ops.conv2d(xd, wd, bd, 256, 3, 3, 1, (1, 1, 1, 1), up_to=(384, 768), gn_stats=True, w_phase=wph)
    torch.cuda.synchronize()
    assert _patch_launches() - before == 4
    assert_close(to_nchw(y), ref, dtype, "256 -> 256 upsampler on the 2x2 patch kernel")
    y2 = ops.conv2d(xd, wd, bd, 256, 3, 3, 1, (1, 1, 1, 1), up_to=(384, 768), gn_stats=True, w_phase=wph)
    assert torch.equal(y, y2) and torch.equal(y._e2eft_gn.partial, y2._e2eft_gn.partial)
