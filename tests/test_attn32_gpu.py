"""Fused attention in strict fp32 (csrc/attn32.hip, round 5): e2eft_attn_fwd / e2eft_attn_fwd_lse / e2eft_attn_bwd with dtype E2EFT_F32, head dim 64 — the attention of the
reference's training recipe (fp32, `--enable_xformers_memory_efficient_attention`: training/scripts/train_marigold_e2e_ft_depth.sh:15,17, training/train.py:308-318;
semantics GeoWizard/geowizard/models/attention.py:338-343,482-497).  Against torch in float64 (F.scaled_dot_product_attention and its autograd), against the GEMM + softmax
form it replaces, on ragged query / key counts, one ... many 64-key tiles, two keys (the empty-prompt cross-attention), fused q|k|v column views, GeoWizard's joint keys."""
import pytest
import torch
import torch.nn.functional as TF

from util import rel_err

pytestmark = pytest.mark.gpu
TOL_FWD, TOL_BWD = 2e-5, 1e-4       # max |err| / max |ref|: fp32 products and accumulation, hardware exp2 (1 ulp), online softmax


def _sp(t, heads):
    return t.reshape(t.shape[0], t.shape[1], heads, 64).transpose(1, 2)


def _ref(q, k, v, heads, scale):
    o = TF.scaled_dot_product_attention(_sp(q.double(), heads), _sp(k.double(), heads), _sp(v.double(), heads), scale=scale)
    return o.transpose(1, 2).reshape(q.shape)


@pytest.mark.parametrize("B,heads,N,Nk", [(2, 5, 144, 144), (1, 2, 300, 300), (3, 1, 128, 129), (2, 3, 576, 2), (1, 5, 200, 77), (2, 2, 96, 193), (1, 4, 64, 1280), (1, 1, 1, 1)])
def test_forward_and_lse_against_float64(dev, B, heads, N, Nk):
    from diffusion_e2e_ft_amd import ops
    g = torch.Generator().manual_seed(N * 3 + Nk)
    C = heads * 64
    q, k, v = (torch.randn(B, n, C, generator=g) for n in (N, Nk, Nk))
    q[0, 0] *= 4.0                                   # a sharp row: the running maximum moves late
    scale = 64 ** -0.5
    o, lse = ops.attention(q.to(dev), k.to(dev), v.to(dev), heads, scale, return_lse=True)
    torch.cuda.synchronize()
    assert o.dtype == torch.float32 and rel_err(o, _ref(q, k, v, heads, scale)) <= TOL_FWD
    s = torch.einsum("bhqd,bhkd->bhqk", _sp(q.double(), heads), _sp(k.double(), heads)) * scale
    want_lse = torch.logsumexp(s, dim=-1) / torch.log(torch.tensor(2.0, dtype=torch.float64))       # base 2, as the 16-bit kernels store it
    assert (lse.double().cpu() - want_lse).abs().max().item() <= 2e-5


def test_fused_qkv_views_and_joint_keys(dev):
    from diffusion_e2e_ft_amd import ops
    g = torch.Generator().manual_seed(5)
    B, heads, N = 2, 2, 320
    C = heads * 64
    qkv = torch.randn(B, N, 3 * C, generator=g)
    d = qkv.to(dev)
    o = ops.attention(d[..., :C], d[..., C:2 * C], d[..., 2 * C:], heads, 0.125)          # row stride 3C: slices of one projection
    assert rel_err(o, _ref(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, 0.125)) <= TOL_FWD
    # GeoWizard joint attention (attention.py:482-491): both halves of the batch attend to the keys of both; 200 keys per segment: tiles span the boundary
    Bh, Nj = 2, 200
    q, k, v = (torch.randn(2 * Bh, Nj, C, generator=g) for _ in range(3))
    o = ops.attention(q.to(dev), k.to(dev), v.to(dev), heads, 0.125, kv_nseg=2, kv_bmod=Bh)
    kj = torch.cat([torch.cat([k[:Bh], k[Bh:]], dim=1)] * 2, dim=0)
    vj = torch.cat([torch.cat([v[:Bh], v[Bh:]], dim=1)] * 2, dim=0)
    assert rel_err(o, _ref(q, kj, vj, heads, 0.125)) <= TOL_FWD


@pytest.mark.parametrize("B,heads,N,Nk", [(2, 5, 144, 144), (1, 2, 300, 300), (2, 1, 130, 65), (1, 3, 64, 2), (1, 2, 96, 77), (1, 1, 257, 448)])
def test_backward_against_float64_autograd(dev, B, heads, N, Nk):
    from diffusion_e2e_ft_amd import ops
    g = torch.Generator().manual_seed(N + 7 * Nk)
    C = heads * 64
    q, k, v = (torch.randn(B, n, C, generator=g) for n in (N, Nk, Nk))
    do = torch.randn(B, N, C, generator=g)
    scale = 64 ** -0.5
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    _ref(qd, kd, vd, heads, scale).backward(do.double())
    dq, dk, dv = (t.to(dev) for t in (q, k, v))
    o, lse = ops.attention(dq, dk, dv, heads, scale, return_lse=True)
    gq, gk, gv = torch.empty_like(dq), torch.empty_like(dk), torch.empty_like(dv)
    ops.attention_bwd(dq, dk, dv, o, do.to(dev), lse, heads, scale, gq, gk, gv)
    torch.cuda.synchronize()
    for name, got, want in (("dq", gq, qd.grad), ("dk", gk, kd.grad), ("dv", gv, vd.grad)):
        assert rel_err(got, want) <= TOL_BWD, (name, rel_err(got, want))


def test_autograd_function_routes_fp32_to_the_fused_kernels_and_agrees_with_the_gemm_form(dev):
    """autograd.attention on packed projections (what modules.Attention calls in training): self-attention qkv [B,N,3C] and cross-attention q + kv; the fused fp32
    path against the GEMM + softmax form of rounds 1-4 (F.FUSED_FP32_ATTENTION = False), forward and all gradients"""
    from diffusion_e2e_ft_amd import autograd as F, ops
    g = torch.Generator().manual_seed(11)
    B, heads, N, L = 2, 5, 200, 77
    C = heads * 64
    for kv_len in (None, L):
        qkv0 = torch.randn(B, N, 3 * C if kv_len is None else C, generator=g).to(dev)
        kv0 = None if kv_len is None else torch.randn(B, kv_len, 2 * C, generator=g).to(dev)
        do = torch.randn(B, N, C, generator=g).to(dev)
        res = {}
        for fused in (True, False):
            F.FUSED_FP32_ATTENTION = fused
            try:
                qkv = qkv0.clone().requires_grad_(True)
                kv = None if kv0 is None else kv0.clone().requires_grad_(True)
                timer = ops.KernelTimer()
                ops.TIMER = timer
                o = F.attention(qkv, kv, heads, 0.125)
                o.backward(do)
                torch.cuda.synchronize()
                ops.TIMER = None
                fams = set(k for k, v_ in timer.summary().items() if v_["launches"])
                res[fused] = (o.detach(), qkv.grad, None if kv is None else kv.grad, fams)
            finally:
                F.FUSED_FP32_ATTENTION = True
                ops.TIMER = None
        assert "attn" in res[True][3] and "attn_bwd" in res[True][3] and "igemm" not in res[True][3], res[True][3]
        assert "igemm" in res[False][3]
        for a, b_ in zip(res[True][:3], res[False][:3]):
            if a is not None:
                assert rel_err(a, b_) <= 1e-4, rel_err(a, b_)


def test_argument_errors(dev):
    import ctypes as C
    from diffusion_e2e_ft_amd import _lib, ops
    x = torch.randn(1, 16, 64 + 2, device=dev)[..., :64]            # row stride 66 floats: not a multiple of 4
    with pytest.raises(RuntimeError, match="row strides"):
        ops.attention(x, x, x, 1, 0.125)
