"""The algebra behind e2eft_upconv2x_fwd, checked on the CPU with torch: `F.interpolate(x, scale_factor=2, mode="nearest")` followed by a 3x3 / pad-1 convolution
(diffusers Upsample2D — the upsamplers of the UNet and of the VAE decoder) equals four 2x2 convolutions of the low-resolution input, one per output parity,
with the phase weights of autograd.phase_conv_weight and pads (1 - py, 1 - px) on the top / left — including the image border, where the zero padding of the
upsampled image coincides with zero padding of the source."""
import torch
import torch.nn.functional as TF


def test_phase_weights_reproduce_upsample_then_conv3x3():
    from diffusion_e2e_ft_amd import autograd as F
    g = torch.Generator().manual_seed(0)
    B, Ci, Co, H, W = 2, 5, 7, 6, 9
    conv = torch.nn.Conv2d(Ci, Co, 3, padding=1).double()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g, dtype=torch.float64))
    x = torch.randn(B, Ci, H, W, generator=g, dtype=torch.float64)
    want = conv(TF.interpolate(x, scale_factor=2.0, mode="nearest"))
    wp = F.phase_conv_weight(conv, torch.float64)                 # [4, Co, (i, j, c)]
    assert tuple(wp.shape) == (4, Co, 4 * Ci)
    got = torch.empty_like(want)
    for py in range(2):
        for px in range(2):
            w = wp[2 * py + px].reshape(Co, 2, 2, Ci).permute(0, 3, 1, 2)      # OIHW 2x2
            xp = TF.pad(x, (1 - px, px, 1 - py, py))                            # left, right, top, bottom: the GEMM pads top / left by 1 - p, the far side is implicit
            got[:, :, py::2, px::2] = TF.conv2d(xp, w, conv.bias)
    assert (got - want).abs().max().item() < 1e-12
    # cached per parameter version
    assert F.phase_conv_weight(conv, torch.float64) is wp
    with torch.no_grad():
        conv.weight.mul_(2.0)
    assert (F.phase_conv_weight(conv, torch.float64) - 2 * wp).abs().max().item() < 1e-12


def test_dgrad_is_one_4x4_stride2_convolution_of_dy():
    """autograd.upconv_dgrad_weight: d(upsample -> conv3x3) / d(low-resolution input) = conv(dY, 4x4, stride 2, pad 1) — against torch autograd in float64"""
    from diffusion_e2e_ft_amd import autograd as F
    g = torch.Generator().manual_seed(1)
    B, Ci, Co, H, W = 2, 8, 16, 5, 7                    # (channel counts that need no padding in float64: epc = 2)
    conv = torch.nn.Conv2d(Ci, Co, 3, padding=1).double()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g, dtype=torch.float64))
    x = torch.randn(B, Ci, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(B, Co, 2 * H, 2 * W, generator=g, dtype=torch.float64)
    conv(TF.interpolate(x, scale_factor=2.0, mode="nearest")).backward(dy)
    wt = F.upconv_dgrad_weight(conv, torch.float64)                       # [Ci, (u, v, co)]
    assert tuple(wt.shape) == (Ci, 16 * Co)
    w4 = wt.reshape(Ci, 4, 4, Co).permute(0, 3, 1, 2)                     # as an OIHW weight: out channels = ci, in channels = co
    got = TF.conv2d(dy, w4, None, stride=2, padding=1)
    assert got.shape == x.shape and (got - x.grad).abs().max().item() < 1e-11
