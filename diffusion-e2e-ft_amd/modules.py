"""Leaf modules and blocks of the SD-v2 UNet / SD VAE with diffusers-compatible parameter names, computing through
libe2eft (HIP) only.  Internally every activation is NHWC `[B, H, W, C]` (tokens `[B, H*W, C]` are the same memory).

The reference composes these blocks in GeoWizard/geowizard/models/unet_2d_blocks.py, transformer_2d.py, attention.py
(vendored twins of diffusers); leaf semantics are diffusers==0.30.2 (not in the reference tree).  Citations per class.
"""

import torch
from torch import nn

from . import ops
from . import autograd as F


# ------------------------------------------------------------------------------------------------------------
# derived tensors (packed conv weights, fused QKV matrices, casts, transposes) are cached per parameter version in autograd.py
_cached = F.cached


def packed_conv_weight(conv):
    """[Co,Ci,kh,kw] -> OHWI rows [Co, kh*kw*Ci_pad] (Ci padded with zeros to a 16-byte multiple)."""
    return F.packed_conv_weight(conv, conv.weight.dtype)


def conv_nhwc(conv, x, x2=None, up_to=None, rowadd=None, residual=None, alpha=1.0, pad=None, out=None, gn_stats=True, norm=None):
    """Run any nn.Conv2d-shaped module (weight/bias/stride/padding) on NHWC input through the implicit-GEMM kernel.
    Works for plain torch.nn.Conv2d objects too (training/util/unet_prep.py:6-20 swaps conv_in for one).  Differentiable:
    under autograd the backward runs e2eft_conv2d_dgrad / the wgrad GEMM (autograd.py)."""
    assert out is None
    # gn_stats: nearly every conv output of the path is consumed by a GroupNorm next; the epilogue then emits its statistics
    # norm = (GroupNorm module, silu): conv(norm(x)); inference may never materialise norm(x) (ops.conv2d)
    return F.conv(conv, x, x2=x2, up_to=up_to, rowadd=rowadd, residual=residual, alpha=alpha, pad=pad, gn_stats=gn_stats, norm=norm)


def checkpointed(enabled, fn, *args):
    """Activation recompute of one block (the reference: `torch.utils.checkpoint.checkpoint(create_custom_forward(resnet), ...)` inside
    every UNet block when `unet.enable_gradient_checkpointing()` was called, unet_2d_blocks.py:1136-1161, training/train.py:342-343).
    Non-reentrant torch checkpointing over the libe2eft autograd Functions: nothing the block saved for its backward is kept, the block's
    forward kernels run again when its backward is reached; the recompute is deterministic, so gradients are bit-equal.  GroupNorm
    statistics ride on the tensors as attributes and survive (the block's input / output objects are the same ones)."""
    if enabled and torch.is_grad_enabled() and any(isinstance(a, torch.Tensor) and a.requires_grad for a in args):
        from torch.utils.checkpoint import checkpoint
        return checkpoint(fn, *args, use_reentrant=False)
    return fn(*args)


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
    return F.nchw_to_nhwc(x, cpad=x.shape[1])


def to_nchw_view(y):
    """NHWC [B,H,W,C] -> logical NCHW view (channels_last strides), no copy."""
    return y.permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------------------------
class TimeCond:
    """SiLU(time embedding) plus, in inference, the `time_emb_proj` outputs of every ResnetBlock2D computed by ONE GEMM (unet.py::_batch_small_gemms): the
    blocks receive it as the ARGUMENT they receive the embedding in and pick their slice by identity — per-call state travels with the call (round 3 parked
    the slices in the modules' __dict__: not re-entrant)."""
    __slots__ = ("act", "rows")

    def __init__(self, act, rows=None):
        self.act, self.rows = act, (rows or {})


class CtxCond:
    """encoder_hidden_states plus the k | v projections of it for every cross-attention layer from one GEMM (same idea as TimeCond).
    shared: every image of the batch attends to the SAME context (the pipelines hand the one empty-prompt embedding to the UNet as a stride-0 expand);
    `src` is then that one context [1, L, X] as the caller holds it (a stable tensor: the key of Attention._fold's cache)."""
    __slots__ = ("ctx", "kv", "shared", "src")

    def __init__(self, ctx, kv=None, shared=False, src=None):
        self.ctx, self.kv, self.shared, self.src = ctx, (kv or {}), shared, src


CROSS_ATTN_FOLD = True     # tests / A-B: False keeps the two-token cross-attention on q-projection + attention kernel + out-projection


class Conv2d(nn.Conv2d):
    @ops.device_scoped
    def forward(self, x):  # public NCHW-logical surface (vae.quant_conv(h) etc.)
        return to_nchw_view(conv_nhwc(self, to_nhwc(x)))


class Linear(nn.Linear):
    @ops.device_scoped
    def forward(self, x, residual=None, gn_rows_per_image=0):
        # K that is not a 16-byte multiple (GeoWizard's 10-dim class-embedding input) is zero padded inside F.linear
        return F.linear(x, self.weight, self.bias, residual=residual, owner=self, gn_rows_per_image=gn_rows_per_image)


class GroupNorm(nn.GroupNorm):
    def nhwc(self, x, x2=None, silu=False, split=False):
        """split=True -> (y, x_skip): x_skip is x for the residual path (its gradient is folded into this norm's backward)"""
        return F.groupnorm(x, self.weight, self.bias, self.num_groups, self.eps, silu=silu, x2=x2, split=split)

    @ops.device_scoped
    def forward(self, x):
        return to_nchw_view(self.nhwc(to_nhwc(x)))


class LayerNorm(nn.LayerNorm):
    @ops.device_scoped
    def forward(self, x):
        return F.layernorm(x, self.weight, self.bias, self.eps)


class TimestepEmbedding(nn.Module):
    """diffusers TimestepEmbedding: Linear -> SiLU -> Linear (unet_2d_condition.py:317-323,366-378)."""

    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = Linear(in_dim, dim)
        self.linear_2 = Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


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
        rowadd = None
        if isinstance(temb_act, TimeCond):                   # inference: projected for all blocks at once (unet.py::_batch_small_gemms)
            rowadd = None if torch.is_grad_enabled() else temb_act.rows.get(id(self))     # never under autograd: the slice carries no graph
            temb_act = temb_act.act
        if rowadd is None and self.time_emb_proj is not None:
            rowadd = self.time_emb_proj(temb_act)
        if not torch.is_grad_enabled() and x2 is None:
            # inference: conv(SiLU(GroupNorm(.))) as ONE op — where the library can, the normalised tensor never exists (ops.conv2d, norm=)
            h = conv_nhwc(self.conv1, x, rowadd=rowadd, norm=(self.norm1, True))
            sc = conv_nhwc(self.conv_shortcut, x) if self.conv_shortcut is not None else x
            return conv_nhwc(self.conv2, h, residual=sc, norm=(self.norm2, True))
        # fp32 with a frozen convolution (the VAE under the fp32 recipe): norm + conv as one differentiable op on f16 split planes (autograd._NormConvSplitFn)
        fused = F.norm_conv_split(self.conv1, self.norm1, x, True, split=True) if (x2 is None and rowadd is None) else None
        if fused is not None:
            h, x = fused
        else:
            if x2 is None:
                h, x = self.norm1.nhwc(x, silu=True, split=True)     # x: the same tensor, routed through the norm for its gradient
            else:
                h = self.norm1.nhwc(x, x2=x2, silu=True)
            h = conv_nhwc(self.conv1, h, rowadd=rowadd)
        if self.conv_shortcut is not None:
            sc = conv_nhwc(self.conv_shortcut, x, x2=x2)
        else:
            assert x2 is None
            sc = x
        fused = F.norm_conv_split(self.conv2, self.norm2, h, True, residual=sc)
        if fused is not None:
            return fused
        h = self.norm2.nhwc(h, silu=True)
        return conv_nhwc(self.conv2, h, residual=sc)

    def forward(self, x, temb=None):
        return to_nchw_view(self.nhwc(to_nhwc(x), None if temb is None else F.silu(temb)))


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

    def forward(self, x, ctx=None, residual=None):
        """x [B,N,C] (already normalised), ctx [B,L,X] or None; returns to_out(attn) + residual."""
        B, N, C = x.shape
        d = C // self.heads
        out = self.to_out[0]
        kv_pre = None
        shared_src = None
        if isinstance(ctx, CtxCond):
            kv_pre = ctx.kv.get(id(self))
            shared_src = ctx.src if ctx.shared else None
            ctx = ctx.ctx
        if F.needs_grad(x, ctx, self.to_q.weight, self.to_k.weight, self.to_v.weight, out.weight):
            return out(self._forward_train(x, ctx), residual=residual)
        if (CROSS_ATTN_FOLD and shared_src is not None and shared_src.shape[1] == 2 and not self.joint and self.to_q.bias is None and self.to_k.bias is None
                and self.to_v.bias is None and self.heads <= 64):
            return self._folded(x, shared_src, residual)
        fused = F.fused_attention_ok(x.dtype, d)
        dt = x.dtype
        if fused:
            if ctx is None:
                qkv = F.linear(x, (self.to_q.weight, self.to_k.weight, self.to_v.weight), owner=self, name="wqkv")
                q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
            else:
                q = self.to_q(x)
                kv = kv_pre                                     # inference: keys / values of the (shared) context for all layers at once
                if kv is None or torch.is_grad_enabled():
                    kv = F.linear(ctx, (self.to_k.weight, self.to_v.weight), owner=self, name="wkv")
                k, v = kv[..., :C], kv[..., C:]
            a = ops.attention(q, k, v, self.heads, self.scale, kv_nseg=2 if self.joint else 1, kv_bmod=B // 2 if self.joint else B)
        else:
            src = x if ctx is None else ctx
            cw = lambda lin: F._cat_weight(lin, "w", (lin.weight,), dt, lin.weight.shape[1])
            a = attention_unfused(x, src, cw(self.to_q), F._vec(self.to_q.bias, dt), cw(self.to_k), F._vec(self.to_k.bias, dt), cw(self.to_v),
                                  F._vec(self.to_v.bias, dt), self.heads, self.scale, joint=self.joint)
        return out(a, residual=residual)

    def _fold(self, src, dt):
        """Cross-attention to TWO context tokens shared by the whole batch (the empty prompt of the E2E-FT path: `<|startoftext|><|endoftext|>`,
        marigold_pipeline.py:356-369; attention.py:338-343 is then softmax over two keys) collapses algebraically.  Per head h, with k_j / v_j the projected tokens:
            p1 = softmax([s1, s2])[0] = sigmoid(s1 - s2) = sigmoid(x . g_h),   g_h = scale * Wq[h]^T (k1 - k2)_h            (a C-vector per head)
            attn_h = v2_h + p1 (v1 - v2)_h,   to_out(attn) = c0 + sum_h p1_h Delta_h,   Delta_h = Wo[:, h] (v1 - v2)_h,  c0 = Wo v2 + b_o
        i.e. two GEMMs whose inner / outer dimension is the number of HEADS (5 ... 20) instead of C (320 ... 1280) and a sigmoid: no q projection, no attention
        launch, no out projection — the activation is read once and written once.  -> (G [Hp, C], DeltaT [C, Hp], c0 [C]) in `dt`, built in fp32 once per
        (context, weights) and cached; Hp = heads padded to a 16-byte multiple with zero rows (sigmoid(0) = 0.5 times a zero row of Delta)."""
        out = self.to_out[0]
        ws = [w for w in (self.to_q.weight, self.to_k.weight, self.to_v.weight, out.weight, out.bias) if w is not None]
        key = (src.data_ptr(), src._version, tuple(src.shape), str(src.device), dt, F._key(*ws))
        hit = self.__dict__.get("_fold_cache")
        if hit is not None and hit[0] == key:
            return hit[1]
        with torch.no_grad():
            # fp32 through the library's own exact-fp32 MFMA GEMM (ops.gemm): no vendor BLAS anywhere in an inference process
            h, C = self.heads, self.to_q.weight.shape[0]
            d = C // h
            f32 = lambda w: w.detach().float().contiguous()
            wq, wk, wv, wo = f32(self.to_q.weight), f32(self.to_k.weight), f32(self.to_v.weight), f32(out.weight)
            c32 = f32(src[0])                                                        # [2, X]
            k = ops.gemm(c32, wk)                                                    # [2, C]
            v = ops.gemm(c32, wv)
            # per-head contractions as ONE GEMM each: row h of a block-diagonal [h, C] matrix carries (k1 - k2)_h / (v1 - v2)_h in head h's columns
            ar = torch.arange(h, device=k.device)
            Mk = torch.zeros((h, h, d), dtype=torch.float32, device=k.device)
            Mv = torch.zeros((h, h, d), dtype=torch.float32, device=k.device)
            Mk[ar, ar] = (k[0] - k[1]).view(h, d) * self.scale
            Mv[ar, ar] = (v[0] - v[1]).view(h, d)
            G = ops.gemm(Mk.view(h, C), wq.t().contiguous())                         # [h, C]: G[h, c] = scale * sum_d (k1 - k2)[h, d] Wq[(h, d), c]
            delta = ops.gemm(Mv.view(h, C), wo)                                      # [h, C]: Delta[h, c] = sum_d Wo[c, (h, d)] (v1 - v2)[h, d]
            c0 = ops.gemm(v[1:2].contiguous(), wo, f32(out.bias) if out.bias is not None else None)[0]
            hp = ops.round_up(h, ops.epc(dt))
            Gp = torch.zeros((hp, C), dtype=torch.float32, device=G.device)
            Gp[:h] = G
            Dt = torch.zeros((C, hp), dtype=torch.float32, device=G.device)
            Dt[:, :h] = delta.t()
            val = (Gp.to(dt).contiguous(), Dt.to(dt).contiguous(), c0.to(dt).contiguous())
        # the entry keeps `src` ALIVE: the key holds its address, and the caching allocator hands a freed address to the next tensor of that size
        # (two different fresh [1, 2, X] contexts on successive calls would otherwise hit a stale entry)
        self.__dict__["_fold_cache"] = (key, val, src)
        return val

    def _folded(self, x, src, residual):
        G, Dt, c0 = self._fold(src, x.dtype)
        p = ops.activation(ops.linear(x, G), "sigmoid")                              # [B, N, Hp]
        return ops.linear(p, Dt, c0, residual=residual)

    def _forward_train(self, x, ctx):
        """differentiable path: fused projections -> attention core (autograd.py _AttentionFn)"""
        B, N, C = x.shape
        if ctx is None:
            bias = None if self.to_q.bias is None else torch.cat([self.to_q.bias, self.to_k.bias, self.to_v.bias])
            qkv = F.linear(x, (self.to_q.weight, self.to_k.weight, self.to_v.weight), bias, owner=self, name="wqkv")
            if self.joint:
                # GeoWizard cross-domain self-attention (attention.py:482-491): keys / values of both domains side by side
                Bv = B // 2
                q, kv = qkv[..., :C], qkv[..., C:]
                kv = torch.cat([kv[:Bv], kv[Bv:]], dim=1).repeat(2, 1, 1)
                return F.attention(q.contiguous(), kv, self.heads, self.scale)
            return F.attention(qkv, None, self.heads, self.scale)
        q = self.to_q(x)
        bias = None if self.to_k.bias is None else torch.cat([self.to_k.bias, self.to_v.bias])
        kv = F.linear(ctx, (self.to_k.weight, self.to_v.weight), bias, owner=self, name="wkv")
        return F.attention(q, kv, self.heads, self.scale)


class GEGLU(nn.Module):
    """diffusers GEGLU: proj(x).chunk(2) -> value * gelu_erf(gate) (attention.py:755)."""

    def __init__(self, dim, inner):
        super().__init__()
        self.proj = Linear(dim, 2 * inner)

    def forward(self, x):
        return F.geglu(self.proj(x))


class FeedForward(nn.Module):
    """FeedForward (attention.py:719-777): GEGLU(dim, 4 dim) -> Dropout(0) -> Linear(4 dim, dim)."""

    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, 4 * dim), nn.Dropout(0.0), Linear(4 * dim, dim)])

    def forward(self, x, residual=None):
        return self.net[2](self.net[0](x), residual=residual)


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
        h, x = self.norm.nhwc(x, split=True)
        h = self.proj_in(h.view(B, H * W, C))
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        out = self.proj_out(h, residual=x.reshape(B, H * W, C), gn_rows_per_image=H * W)
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
        B, H, W, C = x.shape
        n, x = self.group_norm.nhwc(x, split=True)
        n = n.view(B, H * W, C)
        out = self.to_out[0]
        if F.needs_grad(x, self.to_q.weight):
            bias = torch.cat([self.to_q.bias, self.to_k.bias, self.to_v.bias])
            qkv = F.linear(n, (self.to_q.weight, self.to_k.weight, self.to_v.weight), bias, owner=self, name="wqkv")
            a = F.attention(qkv, None, 1, C ** -0.5)
            return out(a, residual=x.reshape(B, H * W, C)).view(B, H, W, C)
        dt = x.dtype
        if dt != torch.float32 and C == 512:
            # 16-bit inference: one fused q|k|v projection, then the fused d = 512 kernel (csrc/attn512.hip) — no [B, HW, HW] score matrix
            bias = F.cached(self, "b_qkv_%s" % dt, (self.to_q.bias, self.to_k.bias, self.to_v.bias),
                            lambda: torch.cat([self.to_q.bias.detach().to(dt), self.to_k.bias.detach().to(dt), self.to_v.bias.detach().to(dt)]))
            qkv = F.linear(n, (self.to_q.weight, self.to_k.weight, self.to_v.weight), bias, owner=self, name="wqkv")
            a = ops.attention512(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], C ** -0.5)
        else:
            a = attention_unfused(n, n, self.to_q.weight, F._vec(self.to_q.bias, dt), self.to_k.weight, F._vec(self.to_k.bias, dt), self.to_v.weight,
                                  F._vec(self.to_v.bias, dt), 1, C ** -0.5)
        o = out(a, residual=x.reshape(B, H * W, C), gn_rows_per_image=H * W)
        res = o.view(B, H, W, C)
        st = getattr(o, "_e2eft_gn", None)
        if st is not None:
            res._e2eft_gn = st
        return res
