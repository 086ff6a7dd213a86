"""Leaf modules and blocks of the SD-v2 UNet / SD VAE with diffusers-compatible parameter names, computing through
libe2eft (HIP) only.  Internally every activation is NHWC `[B, H, W, C]` (tokens `[B, H*W, C]` are the same memory).

The reference composes these blocks in GeoWizard/geowizard/models/unet_2d_blocks.py, transformer_2d.py, attention.py
(vendored twins of diffusers); leaf semantics are diffusers==0.30.2 (not in the reference tree).  Citations per class.
"""
import math

import torch
from torch import nn

from . import ops


# ------------------------------------------------------------------------------------------------------------
# derived-tensor cache (packed conv weights, fused QKV matrices): rebuilt whenever a source parameter changes
def _key(*params):
    return tuple((p._version, p.data_ptr(), p.dtype, str(p.device)) for p in params)


def _cached(module, name, params, builder):
    cache = module.__dict__.setdefault("_e2eft_cache", {})
    k = _key(*params)
    hit = cache.get(name)
    if hit is None or hit[0] != k:
        with torch.no_grad():
            hit = (k, builder())
        cache[name] = hit
    return hit[1]


def packed_conv_weight(conv):
    """[Co,Ci,kh,kw] -> OHWI rows [Co, kh*kw*Ci_pad] (Ci padded with zeros to a 16-byte multiple)."""
    w = conv.weight

    def build():
        Co, Ci, kh, kw = w.shape
        e = ops.epc(w.dtype)
        cp = ops.round_up(Ci, e)
        t = w.detach().permute(0, 2, 3, 1)
        if cp != Ci:
            t = torch.nn.functional.pad(t, (0, cp - Ci))
        return t.reshape(Co, kh * kw * cp).contiguous()

    return _cached(conv, "packed", (w,), build)


def conv_nhwc(conv, x, x2=None, up_to=None, rowadd=None, residual=None, alpha=1.0, pad=None, out=None, gn_stats=True):
    """Run any nn.Conv2d-shaped module (weight/bias/stride/padding) on NHWC input through the implicit-GEMM kernel.
    Works for plain torch.nn.Conv2d objects too (training/util/unet_prep.py:6-20 swaps conv_in for one)."""
    _no_grad_guard(conv.weight, x)
    kh, kw = conv.weight.shape[2:]
    stride = conv.stride[0] if isinstance(conv.stride, tuple) else conv.stride
    if pad is None:
        p = conv.padding[0] if isinstance(conv.padding, tuple) else conv.padding
        pad = (p, p, p, p)
    if x2 is None:
        x = ops.pad_channels(x)
    # gn_stats: nearly every conv output of the path is consumed by a GroupNorm next; the epilogue then emits its statistics
    return ops.conv2d(x, packed_conv_weight(conv), conv.bias, conv.weight.shape[0], kh, kw, stride, pad, x2=x2, up_to=up_to,
                      rowadd=rowadd, residual=residual, alpha=alpha, out=out, gn_stats=gn_stats)


def _no_grad_guard(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError(
            "libe2eft round 1 implements the forward (inference) path only; run under torch.no_grad(). "
            "The E2E-FT backward kernels are not built yet (DESIGN.md, 'next').")


def to_nhwc(x):
    """Logical NCHW tensor -> NHWC [B,H,W,C] (zero-copy when x is channels_last with a 16-byte-multiple C)."""
    assert x.dim() == 4
    v = x.permute(0, 2, 3, 1)
    try:
        ops._nhwc_ld(v)
        dense = True
    except ValueError:
        dense = False
    if dense:
        return v
    return ops.nchw_to_nhwc(x.contiguous(), cpad=x.shape[1])


def to_nchw_view(y):
    """NHWC [B,H,W,C] -> logical NCHW view (channels_last strides), no copy."""
    return y.permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------------------------
class Conv2d(nn.Conv2d):
    def forward(self, x):  # public NCHW-logical surface (vae.quant_conv(h) etc.)
        return to_nchw_view(conv_nhwc(self, to_nhwc(x)))


class Linear(nn.Linear):
    def forward(self, x):
        _no_grad_guard(self.weight, x)
        K = self.in_features
        e = ops.epc(self.weight.dtype)
        if K % e != 0:  # e.g. GeoWizard's 10-dim class-embedding input: zero-pad K to a 16-byte multiple
            kp = ops.round_up(K, e)
            w = _cached(self, "wpad", (self.weight,), lambda: torch.nn.functional.pad(self.weight.detach(), (0, kp - K)).contiguous())
            return ops.linear(torch.nn.functional.pad(x, (0, kp - K)), w, self.bias)
        return ops.linear(x, self.weight, self.bias)


class GroupNorm(nn.GroupNorm):
    def nhwc(self, x, x2=None, silu=False):
        _no_grad_guard(self.weight, x)
        return ops.groupnorm(x, self.weight, self.bias, self.num_groups, self.eps, silu=silu, x2=x2)

    def forward(self, x):
        return to_nchw_view(self.nhwc(to_nhwc(x)))


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        _no_grad_guard(self.weight, x)
        return ops.layernorm(x, self.weight, self.bias, self.eps)


class TimestepEmbedding(nn.Module):
    """diffusers TimestepEmbedding: Linear -> SiLU -> Linear (unet_2d_condition.py:317-323,366-378)."""

    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = Linear(in_dim, dim)
        self.linear_2 = Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(ops.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    """diffusers ResnetBlock2D: GN -> SiLU -> conv3x3 (+ time_emb_proj(SiLU(temb))) -> GN -> SiLU -> conv3x3, + (1x1) shortcut.
    Built at unet_2d_blocks.py:1064,1211,2242,2400,667 (UNet) and :1301,2503 (VAE, temb_channels=None)."""

    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5, output_scale_factor=1.0):
        super().__init__()
        self.norm1 = GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = Conv2d(in_channels, out_channels, 3, 1, 1)
        self.time_emb_proj = Linear(temb_channels, out_channels) if temb_channels else None
        self.norm2 = GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = Conv2d(out_channels, out_channels, 3, 1, 1)
        self.conv_shortcut = Conv2d(in_channels, out_channels, 1, 1, 0) if in_channels != out_channels else None
        if output_scale_factor != 1.0:
            raise NotImplementedError("output_scale_factor != 1 is not used by SD-v2 / SD-VAE")

    def nhwc(self, x, temb_act=None, x2=None):
        """x2: second source of a fused channel concat (skip connection, unet_2d_blocks.py:2328,2456)."""
        h = self.norm1.nhwc(x, x2=x2, silu=True)
        rowadd = self.time_emb_proj(temb_act) if self.time_emb_proj is not None else None
        h = conv_nhwc(self.conv1, h, rowadd=rowadd)
        h = self.norm2.nhwc(h, silu=True)
        if self.conv_shortcut is not None:
            sc = conv_nhwc(self.conv_shortcut, x, x2=x2)
        else:
            assert x2 is None
            sc = x
        return conv_nhwc(self.conv2, h, residual=sc)

    def forward(self, x, temb=None):
        return to_nchw_view(self.nhwc(to_nhwc(x), None if temb is None else ops.silu(temb)))


class Downsample2D(nn.Module):
    """conv3x3 stride 2; UNet: padding 1 (unet_2d_blocks.py:1109,1230); VAE: F.pad(0,1,0,1) + padding 0."""

    def __init__(self, channels, padding):
        super().__init__()
        self.conv = Conv2d(channels, channels, 3, 2, padding)
        self.asym = padding == 0

    def nhwc(self, x):
        return conv_nhwc(self.conv, x, pad=(0, 1, 0, 1) if self.asym else None)


class Upsample2D(nn.Module):
    """nearest 2x (or to a forced size) fused into the conv3x3 gather (unet_2d_blocks.py:2285,2417)."""

    def __init__(self, channels):
        super().__init__()
        self.conv = Conv2d(channels, channels, 3, 1, 1)

    def nhwc(self, x, size=None):
        B, H, W, _ = x.shape
        return conv_nhwc(self.conv, x, up_to=(2 * H, 2 * W) if size is None else tuple(size))


# ------------------------------------------------------------------------------------------------------------
def attention_unfused(xq, src, wq, bq, wk, bk, wv, bv, heads, scale, joint=False):
    """softmax(q k^T) v through batched MFMA GEMMs + row softmax (exact-fp32 capable; any head dim).
    xq [B,N,C] queries' input, src [B,Nk,X] keys/values' input.  Returns [B,N,C] (pre to_out)."""
    B, N, C = xq.shape
    Nk, X = src.shape[1], src.shape[2]
    d = C // heads
    dt, dev = xq.dtype, xq.device
    e = ops.epc(dt)
    q = ops.linear(xq, wq, bq)
    k = ops.linear(src, wk, bk)
    nseg = 2 if joint else 1
    nkt = Nk * nseg
    nkp = ops.round_up(nkt, e)
    Bv = B // nseg
    alloc = torch.zeros if nkp != nkt else torch.empty
    vt = alloc((Bv, C, nkp), dtype=dt, device=dev)
    srcc = src.contiguous()
    for seg in range(nseg):
        # V^T[p][c, seg*Nk + j] = sum_x wv[c,x] src[p + seg*Bv][j,x] + bv[c]
        ops.bgemm_raw(dt, C, Nk, X, wv, X, (0, 0), srcc[seg * Bv:], X, (Nk * X, 0), vt[:, :, seg * Nk:], nkp, (C * nkp, 0), Bv, 1,
                      bias=bv, bias_along_m=True)
    s = alloc((B, heads, N, nkp), dtype=dt, device=dev)
    for qh in range(nseg):
        for seg in range(nseg):
            ops.bgemm_raw(dt, N, Nk, d, q[qh * Bv:], C, (N * C, d), k[seg * Bv:], C, (Nk * C, d), s[qh * Bv:, :, :, seg * Nk:], nkp,
                          (heads * N * nkp, N * nkp), Bv, heads)
    ops.softmax_rows_(s.view(-1, nkp), nkt, scale)
    o = torch.empty((B, N, C), dtype=dt, device=dev)
    for qh in range(nseg):
        ops.bgemm_raw(dt, N, d, nkp, s[qh * Bv:], nkp, (heads * N * nkp, N * nkp), vt, nkp, (C * nkp, d * nkp), o[qh * Bv:], C, (N * C, d),
                      Bv, heads)
    return o


class Attention(nn.Module):
    """diffusers Attention (AttnProcessor2_0 semantics): to_q/to_k/to_v (no bias in the UNet), softmax(q k^T / sqrt(d)) v,
    to_out.0 with bias (attention.py:208-217,239-248,338-343,375-380).  joint=True is GeoWizard's cross-domain
    self-attention (XFormersJointAttnProcessor, attention.py:425-513), made unconditional for such checkpoints."""

    def __init__(self, query_dim, heads, cross_attention_dim=None, bias=False, joint=False):
        super().__init__()
        self.heads = heads
        self.scale = (query_dim // heads) ** -0.5
        self.joint = joint
        kv = cross_attention_dim or query_dim
        self.to_q = Linear(query_dim, query_dim, bias=bias)
        self.to_k = Linear(kv, query_dim, bias=bias)
        self.to_v = Linear(kv, query_dim, bias=bias)
        self.to_out = nn.ModuleList([Linear(query_dim, query_dim), nn.Dropout(0.0)])

    def _qkv(self):
        ps = (self.to_q.weight, self.to_k.weight, self.to_v.weight)
        return _cached(self, "wqkv", ps, lambda: torch.cat([p.detach() for p in ps], dim=0).contiguous())

    def _kv(self):
        ps = (self.to_k.weight, self.to_v.weight)
        return _cached(self, "wkv", ps, lambda: torch.cat([p.detach() for p in ps], dim=0).contiguous())

    def forward(self, x, ctx=None, residual=None):
        """x [B,N,C] (already normalised), ctx [B,L,X] or None; returns to_out(attn) + residual."""
        _no_grad_guard(self.to_q.weight, x)
        B, N, C = x.shape
        d = C // self.heads
        fused = x.dtype != torch.float32 and d == 64
        if fused:
            if ctx is None:
                qkv = ops.linear(x, self._qkv())
                q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
            else:
                q = ops.linear(x, self.to_q.weight)
                kv = ops.linear(ctx, self._kv())
                k, v = kv[..., :C], kv[..., C:]
            a = ops.attention(q, k, v, self.heads, self.scale, kv_nseg=2 if self.joint else 1, kv_bmod=B // 2 if self.joint else B)
        else:
            src = x if ctx is None else ctx
            a = attention_unfused(x, src, self.to_q.weight, self.to_q.bias, self.to_k.weight, self.to_k.bias, self.to_v.weight,
                                  self.to_v.bias, self.heads, self.scale, joint=self.joint)
        return ops.linear(a, self.to_out[0].weight, self.to_out[0].bias, residual=residual)


class GEGLU(nn.Module):
    """diffusers GEGLU: proj(x).chunk(2) -> value * gelu_erf(gate) (attention.py:755)."""

    def __init__(self, dim, inner):
        super().__init__()
        self.proj = Linear(dim, 2 * inner)

    def forward(self, x):
        return ops.geglu(self.proj(x))


class FeedForward(nn.Module):
    """FeedForward (attention.py:719-777): GEGLU(dim, 4 dim) -> Dropout(0) -> Linear(4 dim, dim)."""

    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, 4 * dim), nn.Dropout(0.0), Linear(4 * dim, dim)])

    def forward(self, x, residual=None):
        h = self.net[0](x)
        return ops.linear(h, self.net[2].weight, self.net[2].bias, residual=residual)


class BasicTransformerBlock(nn.Module):
    """attention.py:103-413: LN -> self-attn -> +; LN -> cross-attn -> +; LN -> GEGLU FF -> + (residual adds fused in
    the output GEMM epilogues)."""

    def __init__(self, dim, heads, cross_attention_dim, joint=False):
        super().__init__()
        self.norm1 = LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads, joint=joint)
        self.norm2 = LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, heads, cross_attention_dim=cross_attention_dim)
        self.norm3 = LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, h, ctx):
        h = self.attn1(self.norm1(h), None, residual=h)
        h = self.attn2(self.norm2(h), ctx, residual=h)
        return self.ff(self.norm3(h), residual=h)


class Transformer2DModel(nn.Module):
    """transformer_2d.py:147-217,326-423 (continuous input, use_linear_projection): GN(eps 1e-6) -> Linear -> block ->
    Linear -> + residual.  NHWC makes the reference's permutes free."""

    def __init__(self, channels, heads, cross_attention_dim, groups=32, joint=False):
        super().__init__()
        self.norm = GroupNorm(groups, channels, eps=1e-6, affine=True)
        self.proj_in = Linear(channels, channels)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(channels, heads, cross_attention_dim, joint=joint)])
        self.proj_out = Linear(channels, channels)

    def nhwc(self, x, ctx):
        B, H, W, C = x.shape
        h = self.norm.nhwc(x).view(B, H * W, C)
        h = self.proj_in(h)
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        out = ops.linear(h, self.proj_out.weight, self.proj_out.bias, residual=x, gn_rows_per_image=H * W)
        res = out.view(B, H, W, C)
        st = getattr(out, "_e2eft_gn", None)
        if st is not None:
            res._e2eft_gn = st   # the next ResnetBlock2D.norm1 reads this tensor
        return res


class VaeAttention(nn.Module):
    """Attention of the VAE mid block (unet_2d_blocks.py:589-601): GroupNorm(eps 1e-6) on tokens, q/k/v/out Linear WITH
    bias, one head of dim C, residual inside the module."""

    def __init__(self, channels, groups=32):
        super().__init__()
        self.group_norm = GroupNorm(groups, channels, eps=1e-6, affine=True)
        self.to_q = Linear(channels, channels)
        self.to_k = Linear(channels, channels)
        self.to_v = Linear(channels, channels)
        self.to_out = nn.ModuleList([Linear(channels, channels), nn.Dropout(0.0)])

    def nhwc(self, x):
        _no_grad_guard(self.to_q.weight, x)
        B, H, W, C = x.shape
        n = self.group_norm.nhwc(x).view(B, H * W, C)
        a = attention_unfused(n, n, self.to_q.weight, self.to_q.bias, self.to_k.weight, self.to_k.bias, self.to_v.weight,
                              self.to_v.bias, 1, C ** -0.5)
        out = ops.linear(a, self.to_out[0].weight, self.to_out[0].bias, residual=x, gn_rows_per_image=H * W)
        res = out.view(B, H, W, C)
        st = getattr(out, "_e2eft_gn", None)
        if st is not None:
            res._e2eft_gn = st
        return res
