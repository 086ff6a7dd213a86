"""GPU parity of the backward kernels and autograd Functions (include/e2eft.h "Backward pass ...") against torch CPU fp32 autograd
of the same op.  Inputs are quantised to the kernel dtype first, so both sides differentiate the same function; tolerances
are relative to the largest gradient entry (util.TOL, scaled where a gradient is a long fp16/bf16-rounded sum)."""
import math

import pytest
import torch
import torch.nn.functional as TF

from util import DTYPES, TOL, assert_close, q, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(dev):
    from diffusion_e2e_ft_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def F(dev):
    from diffusion_e2e_ft_amd import autograd as _F
    return _F


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _leaf(t, dtype, dev):
    return t.to(dtype).to(dev).requires_grad_(True)


def _ref(t):
    return t.clone().requires_grad_(True)


# ------------------------------------------------------------------------------------------------ data movement
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Z,R,C", [(1, 64, 64), (3, 77, 128), (2, 200, 72), (1, 5, 8), (2, 1000, 320)])
def test_transpose(ops, dev, dtype, Z, R, C):
    x = torch.randn(Z, R, C, generator=_g(R + C)).to(dtype)
    out = ops.transpose(x.to(dev))
    rp = (R + 63) // 64 * 64
    assert out.shape == (Z, C, rp)
    assert torch.equal(out[:, :, :R].cpu(), x.transpose(1, 2))
    assert out[:, :, R:].abs().max().item() == 0 if rp > R else True
    # strided view input (channel slice of a wider buffer), custom padding
    big = torch.randn(Z, R, 3 * C, generator=_g(1)).to(dtype).to(dev)
    e = 4 if dtype == torch.float32 else 8
    rp2 = (R + e - 1) // e * e
    out = ops.transpose(big[..., C:2 * C], rows_pad=rp2)
    assert torch.equal(out[:, :, :R], big[..., C:2 * C].transpose(1, 2))


def _unfold_ref(x, kh, kw, stride, pad, up_to):
    """[B,C,H,W] fp32 -> [kh*kw*C (tap-major, then channel), B*ho*wo]"""
    if up_to is not None:
        x = TF.interpolate(x, size=up_to, mode="nearest")
    pt, pb, pl, pr = pad
    x = TF.pad(x, (pl, pr, pt, pb))
    B, C = x.shape[:2]
    u = TF.unfold(x, (kh, kw), stride=stride)            # [B, C*kh*kw, L], channel-major then taps
    L = u.shape[2]
    u = u.view(B, C, kh * kw, L).permute(2, 1, 0, 3).reshape(kh * kw * C, B * L)
    return u


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    dict(B=2, H=9, W=7, c1=16, c2=0, k=3, s=1, pad=(1, 1, 1, 1), up=None),
    dict(B=1, H=12, W=10, c1=8, c2=24, k=3, s=1, pad=(1, 1, 1, 1), up=None),
    dict(B=2, H=12, W=8, c1=72, c2=0, k=3, s=2, pad=(1, 1, 1, 1), up=None),
    dict(B=2, H=5, W=6, c1=16, c2=0, k=3, s=1, pad=(1, 1, 1, 1), up=(10, 12)),
    dict(B=1, H=5, W=6, c1=8, c2=0, k=3, s=1, pad=(1, 1, 1, 1), up=(9, 11)),
    dict(B=3, H=6, W=6, c1=40, c2=0, k=1, s=1, pad=(0, 0, 0, 0), up=None),
])
def test_im2col_t(ops, dev, dtype, case):
    c = case
    g = _g(c["H"] * 31 + c["c1"])
    x = torch.randn(c["B"], c["c1"] + c["c2"], c["H"], c["W"], generator=g).to(dtype)
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev)
    x1 = xn[..., :c["c1"]]
    x2 = xn[..., c["c1"]:] if c["c2"] else None
    col, P, Pp = ops.im2col_t(x1, x2, c["k"], c["k"], c["s"], c["pad"], c["up"])
    ref = _unfold_ref(x.float(), c["k"], c["k"], c["s"], c["pad"], c["up"])
    assert ref.shape[1] == P
    assert torch.equal(col[:, :P].float().cpu(), ref)
    assert Pp == P or col[:, P:].abs().max().item() == 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_colsum_and_upsample_bwd(ops, dev, dtype):
    g = _g(4)
    x = q(torch.randn(6 * 50, 72, generator=g), dtype)
    out = ops.colsum(x.to(dtype).to(dev), groups=6, alpha=0.5)
    assert_close(out, 0.5 * x.view(6, 50, 72).sum(1), torch.float32, "colsum", scale=10)
    out = ops.colsum(x.to(dtype).to(dev)[:, 8:24], groups=1)
    assert_close(out, x[:, 8:24].sum(0, keepdim=True), torch.float32, "colsum view", scale=10)
    for (H, W, hl, wl) in [(5, 6, 10, 12), (5, 6, 9, 11), (4, 4, 4, 4)]:
        dy = q(torch.randn(2, 16, hl, wl, generator=g), dtype)
        xin = torch.zeros(2, 16, H, W, requires_grad=True)
        TF.interpolate(xin, size=(hl, wl), mode="nearest").backward(dy)
        got = ops.upsample_nearest_bwd(dy.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev), H, W)
        assert_close(got.permute(0, 3, 1, 2), xin.grad, dtype, "upsample_bwd %s" % ((H, W, hl, wl),))


# ------------------------------------------------------------------------------------------------ convolution
class _Conv(torch.nn.Conv2d):
    pass


CONV_CASES = [
    dict(name="3x3", B=2, H=10, W=12, c1=16, c2=0, co=24, k=3, s=1, p=1, up=None),
    dict(name="3x3 wide", B=1, H=16, W=16, c1=128, c2=0, co=64, k=3, s=1, p=1, up=None),
    dict(name="1x1", B=2, H=7, W=9, c1=40, c2=0, co=16, k=1, s=1, p=0, up=None),
    dict(name="s2", B=2, H=12, W=10, c1=16, c2=0, co=16, k=3, s=2, p=1, up=None),
    dict(name="s2 odd", B=1, H=9, W=11, c1=64, c2=0, co=32, k=3, s=2, p=1, up=None),
    dict(name="up2", B=2, H=5, W=6, c1=16, c2=0, co=16, k=3, s=1, p=1, up=(10, 12)),
    dict(name="up forced", B=1, H=5, W=6, c1=16, c2=0, co=8, k=3, s=1, p=1, up=(9, 11)),
    dict(name="concat", B=2, H=8, W=8, c1=24, c2=40, co=32, k=3, s=1, p=1, up=None),
    dict(name="concat 1x1", B=2, H=8, W=8, c1=64, c2=64, co=32, k=1, s=1, p=0, up=None),
    dict(name="conv_out", B=2, H=8, W=8, c1=32, c2=0, co=4, k=3, s=1, p=1, up=None),
    dict(name="vae conv_out", B=1, H=8, W=8, c1=16, c2=0, co=3, k=3, s=1, p=1, up=None),
    dict(name="latent in", B=2, H=8, W=8, c1=4, c2=0, co=32, k=3, s=1, p=1, up=None),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES, ids=[c["name"] for c in CONV_CASES])
def test_conv_function_gradients(F, dev, dtype, case):
    c = case
    g = _g(len(c["name"]) * 13 + c["c1"])
    cin = c["c1"] + c["c2"]
    B, H, W, co = c["B"], c["H"], c["W"], c["co"]
    conv = _Conv(cin, co, c["k"], c["s"], c["p"])
    with torch.no_grad():
        conv.weight.copy_(q(torch.randn(conv.weight.shape, generator=g) / math.sqrt(cin * c["k"] ** 2), dtype))
        conv.bias.copy_(q(torch.randn(co, generator=g), dtype))
    x = q(torch.randn(B, cin, H, W, generator=g), dtype)
    hl, wl = c["up"] if c["up"] else (H, W)
    ho = (hl + 2 * c["p"] - c["k"]) // c["s"] + 1
    wo = (wl + 2 * c["p"] - c["k"]) // c["s"] + 1
    rowadd = q(torch.randn(B, co, generator=g), dtype)
    res = q(torch.randn(B, co, ho, wo, generator=g), dtype)
    dy = q(torch.randn(B, co, ho, wo, generator=g), dtype)
    alpha = 0.5
    # reference
    xr, rr, ar = _ref(x), _ref(res), _ref(rowadd)
    wr, br = _ref(conv.weight.detach()), _ref(conv.bias.detach())
    xi = TF.interpolate(xr, size=c["up"], mode="nearest") if c["up"] else xr
    yr = alpha * (TF.conv2d(xi, wr, br, c["s"], c["p"]) + ar[:, :, None, None]) + rr
    yr.backward(dy)
    # libe2eft
    dconv = _Conv(cin, co, c["k"], c["s"], c["p"]).to(dev).to(dtype)
    with torch.no_grad():
        dconv.weight.copy_(conv.weight)
        dconv.bias.copy_(conv.bias)
    xn = x.permute(0, 2, 3, 1).contiguous()
    x1 = _leaf(xn[..., :c["c1"]].contiguous(), dtype, dev)
    x2 = _leaf(xn[..., c["c1"]:].contiguous(), dtype, dev) if c["c2"] else None
    rd = _leaf(res.permute(0, 2, 3, 1).contiguous(), dtype, dev)
    ad = _leaf(rowadd, dtype, dev)
    y = F.conv(dconv, x1, x2=x2, up_to=c["up"], rowadd=ad, residual=rd, alpha=alpha)
    assert y.grad_fn is not None
    assert_close(y.permute(0, 3, 1, 2), yr, dtype, "fwd")
    y.backward(dy.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev))
    gx = x1.grad if x2 is None else torch.cat([x1.grad, x2.grad], dim=-1)
    assert_close(gx.permute(0, 3, 1, 2), xr.grad, dtype, "dx", scale=2)
    assert_close(dconv.weight.grad, wr.grad, dtype, "dw", scale=4)
    assert_close(dconv.bias.grad, br.grad, dtype, "dbias", scale=4)
    assert_close(ad.grad, ar.grad, dtype, "drowadd", scale=4)
    assert_close(rd.grad.permute(0, 3, 1, 2), rr.grad, dtype, "dres")


def test_conv_fp32_params_with_16bit_activations(F, dev):
    """mixed setup: fp32 master weights, bf16 activations -> gradients come back in fp32 with the parameter's shape"""
    g = _g(2)
    conv = _Conv(16, 16, 3, 1, 1).to(dev)
    x = torch.randn(1, 8, 8, 16, generator=g).to(torch.bfloat16).to(dev).requires_grad_(True)
    y = F.conv(conv, x)
    y.float().sum().backward()
    assert conv.weight.grad.dtype == torch.float32 and conv.weight.grad.shape == conv.weight.shape
    xr = x.detach().float().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    wq = conv.weight.detach().cpu().to(torch.bfloat16).float().requires_grad_(True)
    TF.conv2d(xr, wq, conv.bias.detach().cpu().to(torch.bfloat16).float(), 1, 1).sum().backward()
    assert rel_err(conv.weight.grad, wq.grad) < 2e-2 and rel_err(x.grad.permute(0, 3, 1, 2), xr.grad) < 2e-2


# ------------------------------------------------------------------------------------------------ linear
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,Ns", [((2, 50), 64, (64,)), ((3, 33), 128, (64, 64, 64)), ((2,), 10, (128,)), ((1, 77), 72, (64, 64)), ((4, 9), 320, (200,))])
def test_linear_function_gradients(F, dev, dtype, M, K, Ns):
    g = _g(K + sum(Ns))
    N = sum(Ns)
    ws = [q(torch.randn(n, K, generator=g) / math.sqrt(K), dtype) for n in Ns]
    bias = q(torch.randn(N, generator=g), dtype)
    x = q(torch.randn(*M, K, generator=g), dtype)
    res = q(torch.randn(*M, N, generator=g), dtype)
    dy = q(torch.randn(*M, N, generator=g), dtype)
    xr, rr, br = _ref(x), _ref(res), _ref(bias)
    wr = [_ref(w) for w in ws]
    yr = TF.linear(xr, torch.cat(wr, 0), br) + rr
    yr.backward(dy)
    owner = torch.nn.Module()
    wd = [torch.nn.Parameter(w.to(dtype).to(dev)) for w in ws]
    bd, xd, rd = _leaf(bias, dtype, dev), _leaf(x, dtype, dev), _leaf(res, dtype, dev)
    y = F.linear(xd, tuple(wd), bd, residual=rd, owner=owner, name="wcat")
    assert_close(y, yr, dtype, "fwd")
    y.backward(dy.to(dtype).to(dev))
    assert_close(xd.grad, xr.grad, dtype, "dx", scale=2)
    for a, b in zip(wd, wr):
        assert_close(a.grad, b.grad, dtype, "dw", scale=4)
    assert_close(bd.grad, br.grad, dtype, "dbias", scale=4)
    assert_close(rd.grad, rr.grad, dtype, "dres")


# ------------------------------------------------------------------------------------------------ norms / activations
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,c1,c2,G,silu", [(2, 9, 7, 64, 0, 32, True), (1, 16, 16, 32, 0, 32, False), (2, 8, 8, 24, 40, 32, True),
                                               (3, 5, 5, 128, 0, 32, True), (1, 40, 40, 320, 0, 32, True)])
def test_groupnorm_function_gradients(F, dev, dtype, B, H, W, c1, c2, G, silu):
    g = _g(c1 + c2 + H)
    C = c1 + c2
    x = q(torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3, dtype)
    ga, be = q(1 + 0.2 * torch.randn(C, generator=g), dtype), q(0.2 * torch.randn(C, generator=g), dtype)
    dy = q(torch.randn(B, C, H, W, generator=g), dtype)
    xr, gr, br = _ref(x), _ref(ga), _ref(be)
    yr = TF.group_norm(xr, G, gr, br, 1e-5)
    if silu:
        yr = TF.silu(yr)
    yr.backward(dy)
    xn = x.permute(0, 2, 3, 1).contiguous()
    x1 = _leaf(xn[..., :c1].contiguous(), dtype, dev)
    x2 = _leaf(xn[..., c1:].contiguous(), dtype, dev) if c2 else None
    gd, bd = _leaf(ga, dtype, dev), _leaf(be, dtype, dev)
    y = F.groupnorm(x1, gd, bd, G, 1e-5, silu=silu, x2=x2)
    assert_close(y.permute(0, 3, 1, 2), yr, dtype, "fwd", scale=2)
    y.backward(dy.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev))
    gx = x1.grad if x2 is None else torch.cat([x1.grad, x2.grad], dim=-1)
    assert_close(gx.permute(0, 3, 1, 2), xr.grad, dtype, "dx", scale=3)
    assert_close(gd.grad, gr.grad, dtype, "dgamma", scale=4)
    assert_close(bd.grad, br.grad, dtype, "dbeta", scale=4)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,C", [((2, 37), 320), ((1, 100), 64), ((3, 9), 1280), ((2, 5), 640)])
def test_layernorm_geglu_silu_gradients(F, dev, dtype, rows, C):
    g = _g(C)
    x = q(torch.randn(*rows, C, generator=g) * 2 + 0.5, dtype)
    ga, be = q(1 + 0.2 * torch.randn(C, generator=g), dtype), q(0.2 * torch.randn(C, generator=g), dtype)
    dy = q(torch.randn(*rows, C, generator=g), dtype)
    xr, gr, br = _ref(x), _ref(ga), _ref(be)
    TF.layer_norm(xr, (C,), gr, br, 1e-5).backward(dy)
    xd, gd, bd = _leaf(x, dtype, dev), _leaf(ga, dtype, dev), _leaf(be, dtype, dev)
    F.layernorm(xd, gd, bd, 1e-5).backward(dy.to(dtype).to(dev))
    assert_close(xd.grad, xr.grad, dtype, "ln dx", scale=3)
    assert_close(gd.grad, gr.grad, dtype, "ln dgamma", scale=4)
    assert_close(bd.grad, br.grad, dtype, "ln dbeta", scale=4)
    # GEGLU on [rows, 2C] and SiLU
    h = q(torch.randn(*rows, 2 * C, generator=g), dtype)
    hr = _ref(h)
    (hr[..., :C] * TF.gelu(hr[..., C:])).backward(dy)
    hd = _leaf(h, dtype, dev)
    F.geglu(hd).backward(dy.to(dtype).to(dev))
    assert_close(hd.grad, hr.grad, dtype, "geglu dh", scale=2)
    sr = _ref(x)
    TF.silu(sr).backward(dy)
    sd = _leaf(x, dtype, dev)
    F.silu(sd).backward(dy.to(dtype).to(dev))
    assert_close(sd.grad, sr.grad, dtype, "silu dx", scale=2)


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(q_, k_, v_, heads, scale):
    B, N, C = q_.shape
    d = C // heads
    sp = lambda t: t.view(B, -1, heads, d).transpose(1, 2)
    s = (sp(q_) @ sp(k_).transpose(-1, -2)) * scale
    return (torch.softmax(s, -1) @ sp(v_)).transpose(1, 2).reshape(B, N, C)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,heads,d", [(2, 100, 2, 64), (1, 256, 5, 64), (2, 64, 1, 128), (1, 144, 1, 512), (1, 321, 2, 64)])
def test_self_attention_gradients(F, dev, dtype, B, N, heads, d):
    g = _g(N + d)
    C = heads * d
    qkv = q(torch.randn(B, N, 3 * C, generator=g), dtype)
    do = q(torch.randn(B, N, C, generator=g), dtype)
    r = _ref(qkv)
    _attn_ref(r[..., :C], r[..., C:2 * C], r[..., 2 * C:], heads, d ** -0.5).backward(do)
    x = _leaf(qkv, dtype, dev)
    o = F.attention(x, None, heads, d ** -0.5)
    o.backward(do.to(dtype).to(dev))
    assert_close(x.grad[..., :C], r.grad[..., :C], dtype, "dq", scale=3)
    assert_close(x.grad[..., C:2 * C], r.grad[..., C:2 * C], dtype, "dk", scale=3)
    assert_close(x.grad[..., 2 * C:], r.grad[..., 2 * C:], dtype, "dv", scale=3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,L,heads,d", [(2, 100, 77, 2, 64), (3, 64, 2, 5, 64), (1, 50, 1, 1, 64), (2, 300, 200, 2, 64), (1, 129, 257, 3, 64)])
def test_cross_attention_gradients(F, dev, dtype, B, N, L, heads, d):
    g = _g(N + L)
    C = heads * d
    qq = q(torch.randn(B, N, C, generator=g), dtype)
    kv = q(torch.randn(B, L, 2 * C, generator=g), dtype)
    do = q(torch.randn(B, N, C, generator=g), dtype)
    qr, kr = _ref(qq), _ref(kv)
    _attn_ref(qr, kr[..., :C], kr[..., C:], heads, d ** -0.5).backward(do)
    qd, kd = _leaf(qq, dtype, dev), _leaf(kv, dtype, dev)
    F.attention(qd, kd, heads, d ** -0.5).backward(do.to(dtype).to(dev))
    if L == 1:   # softmax over one key: dq and dk are exactly 0 analytically; the fused backward leaves rounding noise of P (dP - D)
        assert qd.grad.float().abs().max().item() < 4 * TOL[dtype] and kd.grad[..., :C].float().abs().max().item() < 4 * TOL[dtype] * N
        assert_close(kd.grad[..., C:], kr.grad[..., C:], dtype, "dv", scale=3)
    else:
        assert_close(qd.grad, qr.grad, dtype, "dq", scale=3)
        assert_close(kd.grad, kr.grad, dtype, "dkv", scale=3)


def test_fused_attention_backward_equals_gemm_softmax_form(F, dev):
    """the two backward implementations (csrc/attn_bwd.hip vs batched GEMMs + softmax kernels) agree with each other"""
    g = _g(77)
    B, N, heads = 2, 200, 3
    C = heads * 64
    qkv = torch.randn(B, N, 3 * C, generator=g).to(torch.bfloat16)
    do = torch.randn(B, N, C, generator=g).to(torch.bfloat16).to(dev)
    grads = []
    for flash in (True, False):
        F.FLASH_BACKWARD = flash
        try:
            x = qkv.to(dev).requires_grad_(True)
            F.attention(x, None, heads, 0.125).backward(do)
            grads.append(x.grad.float())
        finally:
            F.FLASH_BACKWARD = True
    assert rel_err(grads[0], grads[1]) < 2e-2


# ------------------------------------------------------------------------------------------------ heads / losses / optimizer
@pytest.mark.parametrize("dtype", DTYPES)
def test_head_gradients(F, dev, dtype):
    g = _g(12)
    x = q(torch.randn(2, 3, 20, 24, generator=g) * 0.8, dtype)
    dy1 = torch.randn(2, 1, 20, 24, generator=g)
    dy3 = torch.randn(2, 3, 20, 24, generator=g)
    xr = _ref(x)
    torch.clamp(xr.mean(dim=1, keepdim=True), -1, 1).backward(q(dy1, dtype))
    xd = _leaf(x.permute(0, 2, 3, 1).contiguous(), dtype, dev)
    F.depth_head(xd).backward(dy1.to(dtype).to(dev))
    assert_close(xd.grad.permute(0, 3, 1, 2), xr.grad, dtype, "depth head dx", scale=2)
    xr = _ref(x)
    torch.clamp(xr / (torch.norm(xr, p=2, dim=1, keepdim=True) + 1e-5), -1, 1).backward(q(dy3, dtype))
    xd = _leaf(x.permute(0, 2, 3, 1).contiguous(), dtype, dev)
    F.normal_head(xd, clamp=True).backward(dy3.to(dtype).to(dev))
    assert_close(xd.grad.permute(0, 3, 1, 2), xr.grad, dtype, "normal head dx", scale=3)


def test_loss_gradients_match_reference_fixture(F, dev):
    """d loss / d pred of the HIP loss kernels vs the gradients torch autograd gives through the REFERENCE's loss.py
    (tests/golden/hooks_golden.pt, generated by make_golden.py from /root/reference/training/util/loss.py)."""
    import os
    import golden_cases as gc
    hooks = torch.load(os.path.join(os.path.dirname(__file__), "golden", "hooks_golden.pt"))
    pred, tgt, mask = gc.ssi_inputs()
    p = pred.to(dev).requires_grad_(True)
    loss = F.ssi_loss(p, tgt.to(dev), mask.to(dev))
    assert abs(loss.item() - hooks["ssi_loss"].item()) <= 1e-5 * abs(hooks["ssi_loss"].item())
    (3.0 * loss).backward()
    assert rel_err(p.grad, 3.0 * hooks["ssi_dpred"]) < 2e-4
    n, nt, m3 = gc.angular_inputs()
    # keep away from |dot| = 1 where acos' is unbounded: the fixture uses perturbed, renormalised predictions
    p = n.to(dev).requires_grad_(True)
    loss = F.angular_loss(p, nt.to(dev), m3.to(dev))
    assert abs(loss.item() - hooks["angular_loss"].item()) <= 1e-5 * abs(hooks["angular_loss"].item())
    loss.backward()
    assert rel_err(p.grad, hooks["angular_dpred"]) < 2e-4


def test_ssi_loss_gradient_degenerate_images(F, dev):
    """an image with no valid pixel contributes nothing; det <= 0 (single valid pixel) gives scale = shift = 0 and zero gradient"""
    g = _g(3)
    pred = torch.randn(3, 1, 8, 8, generator=g)
    tgt = torch.randn(3, 1, 8, 8, generator=g)
    mask = torch.rand(3, 1, 8, 8, generator=g) > 0.2
    mask[1] = False
    mask[2] = False
    mask[2, 0, 3, 3] = True
    from oracle import losses_ref
    pr = _ref(pred)
    losses_ref.ssi_loss_ref(pr, tgt, mask).backward()
    p = pred.to(dev).requires_grad_(True)
    F.ssi_loss(p, tgt.to(dev), mask.to(dev)).backward()
    assert rel_err(p.grad, pr.grad) < 1e-4
    assert p.grad[1].abs().max().item() == 0 and p.grad[2].abs().max().item() == 0


def test_skipped_loss_terms_match_reference_guards(F, ops, dev):
    """training/train.py:504 (`if val_mask.any()`) and :548,552 (`if not torch.isnan(loss)`): such a micro-batch contributes loss 0 and
    no gradient; a non-finite gradient norm must leave parameters and Adam moments untouched"""
    g = _g(4)
    mask0 = torch.zeros(2, 1, 8, 8, dtype=torch.bool)
    for fn, C in ((F.ssi_loss, 1), (F.angular_loss, 3)):
        pred = torch.randn(2, C, 8, 8, generator=g)
        tgt = torch.randn(2, C, 8, 8, generator=g)
        p = pred.to(dev).requires_grad_(True)
        loss = fn(p, tgt.to(dev), mask0.to(dev))
        loss.backward()
        assert loss.item() == 0.0 and p.grad.abs().max().item() == 0.0, fn
        pn = pred.clone()
        pn[0, 0, 1, 1] = float("nan")
        p = pn.to(dev).requires_grad_(True)
        loss = fn(p, tgt.to(dev), torch.ones(2, 1, 8, 8, dtype=torch.bool, device=dev))
        loss.backward()
        assert loss.item() == 0.0 and torch.isfinite(p.grad).all() and p.grad.abs().max().item() == 0.0, fn
    n = 1000
    p0 = torch.randn(n, generator=g).to(dev)
    p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    gr = torch.randn(n, generator=g).to(dev)
    gr[7] = float("inf")
    ops.adamw_step_(p, gr, m, v, 3e-3, 0.9, 0.999, 1e-8, 1e-2, 1, grad_sumsq=ops.sumsq(gr), grad_scale=1.0, max_norm=1.0)
    assert torch.equal(p, p0) and m.abs().max().item() == 0 and v.abs().max().item() == 0


def test_flat_adamw_matches_torch(ops, dev):
    g = _g(8)
    n = 10007
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    p = p0.clone().to(dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step in range(1, 6):
        gr = torch.randn(n, generator=g) * (5.0 if step == 2 else 0.01)
        ref.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        gd = gr.to(dev)
        ss = ops.sumsq(gd)
        assert abs(ss.item() - float((gr.double() ** 2).sum())) < 1e-6 * float((gr.double() ** 2).sum())
        ops.adamw_step_(p, gd, m, v, 3e-3, 0.9, 0.999, 1e-8, 1e-2, step, grad_sumsq=ss, grad_scale=1.0, max_norm=1.0)
        assert rel_err(p, ref.detach()) < 2e-6, step
    # cast / accumulate
    x = torch.randn(1000, generator=g)
    y = torch.zeros(1000, dtype=torch.bfloat16, device=dev)
    ops.cast_(x.to(dev), y)
    assert torch.equal(y.cpu(), x.to(torch.bfloat16))
    acc = torch.ones(1000, device=dev)
    ops.cast_(y, acc, mul=2.0, accumulate=True)
    assert torch.allclose(acc.cpu(), 1 + 2 * x.to(torch.bfloat16).float())
