"""csrc/wgrad.hip — weight gradients straight from the NHWC tensors (transpose reads; no dY^T / im2col copies) — against torch CPU autograd of
`F.conv2d` / `F.linear` w.r.t. the weight, and against the transpose + im2col_t + split-K GEMM path it replaces: 3x3 stride 1 / 2, 1x1, two-source
concat with the source switch at a 64-channel chunk, Co that is not a multiple of the 128-row tile (320), an odd number of 64-column chunks (half-empty
last tile), pixel counts that are not multiples of the 64-pixel k-tile or of the split, several images, a Linear with 77 x batch rows."""
import pytest
import torch
import torch.nn.functional as TF

from util import rel_err

pytestmark = pytest.mark.gpu
TOL = {torch.float16: 3e-3, torch.bfloat16: 2e-2, torch.float32: 2e-5}      # fp32 (round 6, wgrad32_kernel): exact fp32 products and sums, another summation order than torch's


def _conv_case(dev, dtype, B, H, W, c1, c2, Co, k, stride, seed):
    from diffusion_e2e_ft_amd import ops
    g = torch.Generator().manual_seed(seed)
    x1 = torch.randn(B, c1, H, W, generator=g).to(dtype)
    x2 = torch.randn(B, c2, H, W, generator=g).to(dtype) if c2 else None
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dy = (torch.randn(B, Co, Ho, Wo, generator=g) * 0.1).to(dtype)
    xin = x1.float() if x2 is None else torch.cat([x1, x2], dim=1).float()
    w = torch.zeros(Co, c1 + c2, k, k, requires_grad=True)
    TF.conv2d(xin, w, None, stride=stride, padding=pad).backward(dy.float())
    want = w.grad.permute(0, 2, 3, 1).reshape(Co, -1)            # OHWI rows
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    got = ops.conv2d_wgrad(nh(dy), nh(x1), None if x2 is None else nh(x2), Co, k, k, stride, (pad, pad, pad, pad), 1.0)
    assert got is not None and got.dtype == torch.float32 and tuple(got.shape) == tuple(want.shape)
    torch.cuda.synchronize()
    return rel_err(got, want)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,H,W,c1,c2,Co,k,stride", [
    (2, 16, 16, 64, 0, 64, 3, 1),          # 512 pixels, one tile
    (1, 24, 40, 128, 0, 320, 3, 1),        # Co = 320 (2.5 row tiles), 960 pixels
    (3, 13, 11, 320, 0, 128, 3, 1),        # 45 chunks (odd: half-empty last column tile), 429 pixels (ragged k-tile)
    (2, 16, 16, 192, 128, 128, 3, 1),      # two sources, switch after chunk 3
    (2, 32, 32, 64, 0, 128, 3, 2),         # stride 2
    (4, 20, 20, 256, 0, 192, 1, 1),        # 1x1
    (1, 72, 72, 64, 0, 64, 3, 1),          # 5184 pixels: several pixel splits
    (2, 24, 24, 128, 0, 640, 1, 1),        # 1x1 with Cout >> Cin: the 256 x 128 tile shape (640 = 2.5 row tiles of 256)
    (2, 16, 16, 64, 64, 320, 1, 1),        # 256 x 128 tiles, two sources, ragged rows (320 = 256 + 64)
])
def test_conv_wgrad_against_autograd(dev, dtype, B, H, W, c1, c2, Co, k, stride):
    e = _conv_case(dev, dtype, B, H, W, c1, c2, Co, k, stride, seed=H * 7 + Co + k)
    assert e <= TOL[dtype], e


def test_linear_wgrad_and_fallbacks(dev):
    from diffusion_e2e_ft_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2 * 77 * 5, 320, generator=g).half()
    dy = (torch.randn(2 * 77 * 5, 640, generator=g) * 0.1).half()
    want = dy.float().t() @ x.float()
    got = ops.linear_wgrad(dy.to(dev), x.to(dev), 0.5)
    assert got is not None and rel_err(got, 0.5 * want) <= 3e-3
    # strided operands (column slices of wider buffers), as the fused q|k|v projection hands them over
    wide = torch.randn(770, 3 * 640, generator=g).half().to(dev)
    got2 = ops.linear_wgrad(wide[:, 640:1280], x.to(dev))
    assert got2 is not None and rel_err(got2, wide[:, 640:1280].float().cpu().t() @ x.float()) <= 3e-3
    # strict fp32 (round 6): the same call on wgrad32_kernel, incl. the strided q | k | v slice
    got3 = ops.linear_wgrad(dy.float().to(dev), x.float().to(dev), 0.5)
    assert got3 is not None and rel_err(got3, 0.5 * want) <= 2e-5
    got4 = ops.linear_wgrad(wide.float()[:, 640:1280], x.float().to(dev))
    assert got4 is not None and rel_err(got4, wide[:, 640:1280].float().cpu().t() @ x.float()) <= 2e-5
    # not this kernel's problems -> None (the caller keeps the GEMM path): K not a multiple of 64
    assert ops.linear_wgrad(dy.to(dev), torch.randn(770, 72).half().to(dev)) is None
    assert ops.conv2d_wgrad(torch.zeros(1, 8, 8, 64, device=dev).half(), torch.zeros(1, 8, 8, 8, device=dev).half(), None, 64, 3, 3, 1, (1, 1, 1, 1), 1.0) is None


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_direct_wgrad_equals_the_gemm_path_through_autograd(dev, dtype):
    """the same conv module's weight gradient through autograd with the direct kernel and with the transpose + im2col_t + GEMM path"""
    from diffusion_e2e_ft_amd import ops
    from diffusion_e2e_ft_amd.modules import Conv2d, conv_nhwc
    torch.manual_seed(0)
    conv = Conv2d(128, 192, 3, 1, 1).to(dev, dtype)
    x = torch.randn(2, 24, 24, 128, device=dev).to(dtype).requires_grad_(True)
    gy = (torch.randn(2, 24, 24, 192, device=dev) * 0.1).to(dtype)
    grads = []
    for direct in (True, False):
        ops.WGRAD_DIRECT = direct
        try:
            conv.weight.grad = None
            conv_nhwc(conv, x).backward(gy)
            grads.append(conv.weight.grad.detach().float().clone())
        finally:
            ops.WGRAD_DIRECT = True
    assert grads[0].is_contiguous() and rel_err(grads[0], grads[1]) <= {torch.float16: 2e-3, torch.bfloat16: 1.5e-2, torch.float32: 2e-5}[dtype]
