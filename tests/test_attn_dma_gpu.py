"""e2eft_attn_fwd with E2EFT_OPT_ATTN_DMA = 1 (csrc/attn.hip, attn_fwd_dma_kernel; the default since round 5): K / V tiles delivered by LDS-DMA into a
two-stage ring instead of through registers (option 0: attn_fwd_kernel, the round-3 kernel).  The arithmetic per (query, key) is the production kernel's, instruction for instruction, so outputs and log-sum-exps must be BIT-identical —
ragged query / key counts (partial last tile, a single key, fewer keys than one tile), 1 ... 7 tiles (ring wrap-around), fused q|k|v column views, the
GeoWizard joint (two-segment) keys, a late dominant key (the deferred rescale), fp16 and bf16 — and equal to torch's SDPA within the usual bar."""
import pytest
import torch
import torch.nn.functional as F

from util import assert_close, q

pytestmark = pytest.mark.gpu


@pytest.fixture
def dma():
    from diffusion_e2e_ft_amd import _lib

    def run(fn):
        with _lib.option(_lib.OPT_ATTN_DMA, 0):
            a = fn()
        with _lib.option(_lib.OPT_ATTN_DMA, 1):
            b = fn()
        torch.cuda.synchronize()
        return a, b
    return run


def _ref(qq, kk, vv, heads):
    B, N, Wd = qq.shape
    sp = lambda t: t.reshape(t.shape[0], t.shape[1], heads, 64).transpose(1, 2).double()
    return F.scaled_dot_product_attention(sp(qq), sp(kk), sp(vv)).transpose(1, 2).reshape(B, N, Wd).float()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,Nk", [(144, 144), (300, 300), (1024, 1024), (576, 2), (144, 77), (200, 1), (130, 64), (128, 129), (96, 193), (257, 448), (64, 4608)])
def test_dma_delivery_is_bit_identical(dev, dma, dtype, N, Nk):
    from diffusion_e2e_ft_amd import ops
    g = torch.Generator().manual_seed(N * 7 + Nk)
    B, heads = 3, 5
    qq, kk, vv = (q(torch.randn(B, n, heads * 64, generator=g), dtype) for n in (N, Nk, Nk))
    dq, dk, dv = (t.to(dtype).to(dev) for t in (qq, kk, vv))
    (o0, l0), (o1, l1) = dma(lambda: ops.attention(dq, dk, dv, heads, 64 ** -0.5, return_lse=True))
    assert torch.equal(o0, o1) and torch.equal(l0, l1), ((o0.float() - o1.float()).abs().max().item(), (l0 - l1).abs().max().item())
    assert_close(o1, _ref(qq, kk, vv, heads), dtype, "attention (LDS-DMA)", scale=1.5)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_dma_fused_qkv_views_spike_and_joint_keys(dev, dma, dtype):
    from diffusion_e2e_ft_amd import ops
    g = torch.Generator().manual_seed(21)
    B, heads, N = 2, 2, 320
    C = heads * 64
    qkv = torch.randn(B, N, 3 * C, generator=g)
    qkv[0, 5, :C] *= 6.0
    qkv[0, 300, C:2 * C] = qkv[0, 5, :C]          # key 300 aligned with query 5: a late dominant key
    d = q(qkv, dtype).to(dtype).to(dev)
    a, b = dma(lambda: ops.attention(d[..., :C], d[..., C:2 * C], d[..., 2 * C:], heads, 64 ** -0.5))
    assert torch.equal(a, b)
    # GeoWizard joint attention: both halves attend to the concatenation of both halves' keys (kv_nseg = 2); 200 keys per segment: a tile spans the segment boundary
    Bh, Nj = 2, 200
    qq, kk, vv = (q(torch.randn(2 * Bh, Nj, C, generator=g), dtype).to(dtype).to(dev) for _ in range(3))
    a, b = dma(lambda: ops.attention(qq, kk, vv, heads, 64 ** -0.5, kv_nseg=2, kv_bmod=Bh))
    assert torch.equal(a, b)
    kj = torch.cat([torch.cat([kk[:Bh], kk[Bh:]], dim=1)] * 2, dim=0)
    vj = torch.cat([torch.cat([vv[:Bh], vv[Bh:]], dim=1)] * 2, dim=0)
    assert_close(b, _ref(qq.float().cpu(), kj.float().cpu(), vj.float().cpu(), heads), dtype, "joint attention (LDS-DMA)", scale=1.5)
