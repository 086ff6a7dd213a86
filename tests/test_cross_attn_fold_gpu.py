"""Cross-attention to the two-token empty prompt, folded (modules.Attention._fold, round 5): with one context [1, 2, X] shared by the whole batch — the E2E-FT path's
`<|startoftext|><|endoftext|>` embedding, marigold_pipeline.py:356-369 — softmax over two keys is a sigmoid of the score difference and `to_out(attention(...))`
collapses to two GEMMs whose inner / outer dimension is the number of heads.  Exact algebra: against torch in float64 (fp32 module: 1e-5), against the q-projection +
attention-kernel + out-projection route it replaces (fp16 / bf16: inside the single-op bar), and through the UNet (stride-0 expanded context vs a materialised repeat)."""
import pytest
import torch

from util import TOL, rel_err

pytestmark = pytest.mark.gpu


def _ref(att, x, ctx1, res):
    h, C = att.heads, x.shape[-1]
    d = C // h
    W = lambda lin: lin.weight.detach().double().cpu()
    xq = x.double().cpu() @ W(att.to_q).t()
    c = ctx1[0].double().cpu()
    k, v = c @ W(att.to_k).t(), c @ W(att.to_v).t()
    B, N, _ = xq.shape
    q = xq.view(B, N, h, d)
    s = torch.einsum("bnhd,lhd->bnhl", q, k.view(2, h, d)) * att.scale
    p = torch.softmax(s, dim=-1)
    a = torch.einsum("bnhl,lhd->bnhd", p, v.view(2, h, d)).reshape(B, N, C)
    out = att.to_out[0]
    return a @ W(out).t() + out.bias.detach().double().cpu() + res.double().cpu()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("heads,C,N,B", [(5, 320, 300, 2), (10, 640, 144, 3), (20, 1280, 64, 1)])
def test_folded_cross_attention_matches_float64_and_the_unfolded_route(dev, dtype, heads, C, N, B):
    from diffusion_e2e_ft_amd import modules as M
    torch.manual_seed(heads)
    att = M.Attention(C, heads=heads, cross_attention_dim=1024).to(dev, dtype).eval()
    g = torch.Generator().manual_seed(C)
    x = torch.randn(B, N, C, generator=g).to(dev, dtype)
    res = torch.randn(B, N, C, generator=g).to(dev, dtype)
    ctx1 = (0.5 * torch.randn(1, 2, 1024, generator=g)).to(dev, dtype)
    want = _ref(att, x.float(), ctx1.float(), res.float()).float()
    with torch.no_grad():
        y = att(x, M.CtxCond(ctx1.expand(B, -1, -1).contiguous(), None, shared=True, src=ctx1), residual=res)
        assert att.__dict__.get("_fold_cache") is not None
        first = att.__dict__["_fold_cache"][1][0].data_ptr()
        y2 = att(x, M.CtxCond(ctx1.expand(B, -1, -1).contiguous(), None, shared=True, src=ctx1), residual=res)
        assert att.__dict__["_fold_cache"][1][0].data_ptr() == first and torch.equal(y, y2)      # cached per (context, weights)
        M.CROSS_ATTN_FOLD = False
        try:
            y0 = att(x, ctx1.expand(B, -1, -1).contiguous(), residual=res)
        finally:
            M.CROSS_ATTN_FOLD = True
    torch.cuda.synchronize()
    tol = 1e-5 if dtype == torch.float32 else TOL[dtype]
    assert rel_err(y, want) <= tol, rel_err(y, want)
    assert rel_err(y0, want) <= (3e-5 if dtype == torch.float32 else TOL[dtype])
    assert rel_err(y, y0) <= 2 * tol
    # a changed context (in place: version bump) or changed weights rebuild the fold
    with torch.no_grad():
        ctx1.mul_(-1.0)
        y3 = att(x, M.CtxCond(ctx1.expand(B, -1, -1).contiguous(), None, shared=True, src=ctx1), residual=res)
    assert rel_err(y3, _ref(att, x.float(), ctx1.float(), res.float()).float()) <= tol


def test_fold_cache_cannot_go_stale_when_a_fresh_context_reuses_the_address(dev):
    """the fold is cached under the context tensor's (address, version, shape); factory-made tensors all have version 0 and the caching allocator hands
    a freed block to the next tensor of that size, so the cache entry has to keep the context ALIVE.  Ten different fresh [1, 2, X] contexts in a row, each
    dropped before the next is made: every output must be the one of ITS context (and no two of the contexts may share an address while the cache holds one)."""
    from diffusion_e2e_ft_amd import modules as M
    torch.manual_seed(0)
    heads, C, N = 5, 320, 64
    att = M.Attention(C, heads=heads, cross_attention_dim=1024).to(dev, torch.float32).eval()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, N, C, generator=g).to(dev)
    res = torch.zeros(1, N, C, device=dev)
    held = None
    for i in range(10):
        c_host = 0.5 * torch.randn(1, 2, 1024, generator=g)
        ctx1 = c_host.to(dev)                                  # fresh tensor, _version 0
        assert held is None or ctx1.data_ptr() != held        # the previous context is still referenced by the cache: its block was not recycled
        with torch.no_grad():
            y = att(x, M.CtxCond(ctx1, None, shared=True, src=ctx1), residual=res)
        torch.cuda.synchronize()
        e = rel_err(y, _ref(att, x, c_host, res).float())
        assert e <= 1e-5, (i, e)
        held = ctx1.data_ptr()
        del ctx1


def test_unet_folds_a_stride0_context_and_not_a_materialised_one(dev):
    """the UNet recognises the shared context by its stride-0 batch dimension (what the pipelines pass); a per-image context of the same values takes the
    attention kernels — the two outputs agree inside the fp16 bar, and the fp32 pair to 1e-5"""
    import golden_cases as gc
    from oracle import config
    from diffusion_e2e_ft_amd import modules as M, ops
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    for dtype, tol in ((torch.float32, 2e-5), (torch.float16, 1e-2)):
        unet = UNet2DConditionModel(**config.TINY_UNET)
        unet.load_state_dict(gc.tiny_unet_sd())
        unet = unet.to(dev, dtype).eval()
        x, ctx = gc.unet_inputs((16, 16))
        x = x.to(dev, dtype)
        ctx1 = ctx[:1, :2].contiguous().to(dev, dtype)
        t = torch.tensor(999, device=dev)
        with torch.no_grad():
            timer = ops.KernelTimer()
            ops.TIMER = timer
            a = unet(x, t, ctx1.expand(2, -1, -1)).sample
            torch.cuda.synchronize()
            ops.TIMER = None
            n_attn_folded = sum(v["launches"] for (name, lab), v in timer.by_label().items() if name == "attn" and str(lab).endswith("Nk2"))
            timer = ops.KernelTimer()
            ops.TIMER = timer
            b = unet(x, t, ctx1.repeat(2, 1, 1)).sample
            torch.cuda.synchronize()
            ops.TIMER = None
            n_attn_plain = sum(v["launches"] for (name, lab), v in timer.by_label().items() if name == "attn" and str(lab).endswith("Nk2"))
        assert rel_err(a, b) <= tol, (dtype, rel_err(a, b))
        if dtype != torch.float32:
            assert n_attn_folded == 0 and n_attn_plain > 0, (n_attn_folded, n_attn_plain)
