"""torch.autograd plumbing of the E2E-FT training step (training/train.py:470-568).

The reference gets its gradients from torch autograd over diffusers modules.  Here every differentiable op of the host
layer is a `torch.autograd.Function` whose forward AND backward are libe2eft (HIP) launches; torch only records the graph,
owns the memory and accumulates `.grad`.  Entry points `conv / linear / groupnorm / layernorm / geglu / silu / attention /
depth_head / normal_head / ssi_loss / angular_loss` pick the Function when a gradient is required and the plain inference
wrapper of ops.py otherwise (so the inference path keeps its fused GroupNorm statistics and pays nothing).

Activations keep the NHWC / token layout of the forward; parameters keep the diffusers layout (conv weights OIHW), their
gradients are returned in the parameter's dtype and shape.  Compute dtype follows the activations: fp32 parameters with 16-bit
activations are cast once per parameter version (cached).
"""
import weakref

import torch

from . import ops


def needs_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


# ------------------------------------------------------------------------------------------------------------
# per-parameter derived tensors (packed / transposed / cast weights): rebuilt when the parameter changes
PARAM_EPOCH = 0   # bumped by the flat-buffer optimizer (training.py), which updates parameters behind torch's version counters


def bump_param_epoch():
    global PARAM_EPOCH
    PARAM_EPOCH += 1


def _key(*params):
    # the epoch only concerns parameters a flat-buffer optimizer updates behind torch's version counters (training.FlatAdamW marks them); everything else — the frozen
    # VAE, the text / image encoders — keeps its derived tensors across optimizer steps
    ep = PARAM_EPOCH if any(getattr(p, "_e2eft_flat", None) is not None for p in params) else -1
    return (ep,) + tuple((p._version, p.data_ptr(), p.dtype, str(p.device)) for p in params)


# ---- compute-dtype SHADOW of a flat master buffer --------------------------------------------------------------------------------------------------------
# training.FlatAdamW keeps every trainable parameter in ONE flat fp32 buffer, convolution weights in the order the kernels read them (OHWI: the parameter is a
# channels_last view).  With 16-bit compute the whole buffer is cast ONCE per parameter state by one e2eft_cast launch into a flat 16-bit twin; packed convolution
# weights, (concatenated) Linear weights, biases and norm parameters are then VIEWS of that twin — no per-tensor cast / permute / cat launches (round 3: ~1100
# `bfloat16_copy`, 214 `direct_copy` and 84 `flip` ATen launches per step, 12 ms).
class FlatShadow:
    """flat: the fp32 master buffer; `register(params, offsets)`: the Parameters whose storage it is.  A twin is current for the key (PARAM_EPOCH, flat._version)
    AND the per-parameter version counters recorded when it was cast: FlatAdamW binds a parameter with `p.data = view`, so p has its OWN version counter —
    `p.copy_()` under no_grad (load_state_dict, an EMA copy_to) writes the flat buffer but bumps neither flat._version nor the epoch.  `shadow_view`
    therefore compares the asking parameter's `_version` with the recorded one and re-casts on a mismatch.  (`p.data.copy_()` bumps no counter anywhere in torch;
    call `bump_param_epoch()` after such a write.)"""

    def __init__(self, flat):
        self.flat, self.twins, self.state, self.pver, self.params = flat, {}, {}, {}, []

    def register(self, params, offsets):
        self.params = [(weakref.ref(p), o) for p, o in zip(params, offsets)]      # weak: no cycle parameter -> tag -> shadow -> parameter

    def _versions(self):
        out = {}
        for r, o in self.params:
            p = r()
            if p is not None:
                out[o] = p._version
        return out

    def twin(self, dtype, force=False):
        key = (PARAM_EPOCH, self.flat._version)          # in-place writes to the flat buffer itself bump its version; the optimizer kernel bumps the epoch
        t = self.twins.get(dtype)
        if t is None:
            t = self.twins[dtype] = torch.empty(self.flat.shape, dtype=dtype, device=self.flat.device)
            self.state[dtype] = None
        if force or self.state[dtype] != key:
            with torch.no_grad(), ops.on_device_of(self.flat):
                ops.cast_(self.flat, t)
            self.state[dtype] = key
            self.pver[dtype] = self._versions()
        return t

    def view_for(self, p, off, dtype):
        t = self.twin(dtype)
        if self.pver[dtype].get(off, p._version) != p._version:      # p was written in place through its own version counter since the cast
            t = self.twin(dtype, force=True)
        return t


FLAT_SHADOW_ENABLED = True    # tests / A-B: False sends every derived tensor through the per-tensor cast / pack / cat path of round 3


def shadow_view(p, dtype):
    """p lives in a flat master buffer (FlatAdamW): the view of the buffer's `dtype` twin with p's shape and strides; else None"""
    tag = getattr(p, "_e2eft_flat", None)
    if tag is None or p.dtype == dtype or not FLAT_SHADOW_ENABLED:
        return None
    sh, off = tag
    if p.data_ptr() != sh.flat.data_ptr() + off * sh.flat.element_size() or p.device != sh.flat.device:
        return None                                      # the parameter was re-bound since (.to(), a deepcopy's stale tag): the lazy per-tensor path serves it
    return sh.view_for(p, off, dtype).as_strided(p.shape, p.stride(), off)


def _is_ohwi(w):
    Co, Ci, kh, kw = w.shape
    return w.dim() == 4 and w.stride() == (kh * kw * Ci, 1, kw * Ci, Ci)


def phase_conv_weight(conv, dtype):
    """[4, Co, 2 * 2 * Ci] weights of the four parity phases of `nearest-2x upsample -> conv 3x3 / pad 1` (include/e2eft.h, e2eft_upconv2x_fwd): phase (py, px),
    source offset (i, j) collects the 3x3 taps that land on that source pixel — rows R(0) = ({0}, {1, 2}), R(1) = ({0, 1}, {2}), columns alike.  Summed in fp32
    from the master weights, rounded once to `dtype`; cached per parameter version like the packed weights."""
    w = conv.weight

    def build():
        w32 = w.detach().to(torch.float64 if w.dtype == torch.float64 else torch.float32)      # [Co, Ci, 3, 3]
        R = (((0,), (1, 2)), ((0, 1), (2,)))
        out = torch.empty((4, w32.shape[0], 2, 2, w32.shape[1]), dtype=w32.dtype, device=w32.device)
        for py in range(2):
            for px in range(2):
                for i in range(2):
                    rows = sum(w32[:, :, ky, :] for ky in R[py][i])                      # [Co, Ci, 3]
                    for j in range(2):
                        out[2 * py + px, :, i, j, :] = sum(rows[:, :, kx] for kx in R[px][j])
        return out.reshape(4, w32.shape[0], 4 * w32.shape[1]).to(dtype).contiguous()

    return cached(conv, "wphase_%s" % dtype, (w,), build)


GRAD_SINK_ENABLED = True      # tests / A-B: False hands every parameter gradient to autograd as a fresh tensor (AccumulateGrad then adds it into the flat buffer)


def grad_sink(*params):
    """FlatAdamW(direct_grads=True): the slots of `params` in the flat fp32 gradient buffer, when THIS backward call is the first writer of their gradients since
    zero_grad() (every .grad is None, nobody claimed a slot) and — for more than one parameter — the slots lie back to back.  Returns (flat, views): `flat` the dense
    fp32 1-D view spanning the slots (what a reduction kernel writes), `views` one fresh view per parameter shaped and strided like it (what backward() returns:
    autograd keeps such a tensor as .grad without a copy — AccumulateGrad's no-other-reference path — so the gradient is born in the exchange buffer).  Else None:
    the caller produces an ordinary tensor and autograd accumulates it."""
    if not GRAD_SINK_ENABLED:
        return None
    tags = [getattr(q, "_e2eft_gslot", None) for q in params]
    if any(t is None for t in tags) or any(q.grad is not None for q in params) or any(t[0] is not tags[0][0] for t in tags):
        return None
    owner = tags[0][0]()
    return None if owner is None else owner._claim([t[1] for t in tags], params)


def cached(owner, name, params, builder):
    cache = owner.__dict__.setdefault("_e2eft_cache", {})
    k = _key(*params)
    hit = cache.get(name)
    if hit is None or hit[0] != k:
        with torch.no_grad():
            hit = (k, builder())
        cache[name] = hit
    return hit[1]


def _vec(p, dtype):
    """bias / gamma / beta in the compute dtype"""
    if p is None:
        return None
    if p.dtype == dtype:
        return p.detach()
    sv = shadow_view(p, dtype)           # (asked of the Parameter itself: the flat-buffer tag is a Python attribute, a detached alias does not carry it)
    return sv if sv is not None else p.detach().to(dtype)


def packed_conv_weight(conv, dtype):
    """[Co,Ci,kh,kw] -> OHWI rows [Co, kh*kw*Ci_pad] in `dtype` (Ci padded with zeros to a 16-byte multiple)."""
    w = conv.weight
    if w.dtype != dtype and w.shape[1] % ops.epc(dtype) == 0:
        if _is_ohwi(w):
            sv = shadow_view(w, dtype)
            if sv is not None:                           # the master copy already lies in OHWI order: the packed weight IS the twin's slice
                return sv.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
        elif w.shape[2] * w.shape[3] == 1 and w.is_contiguous():
            sv = shadow_view(w, dtype)
            if sv is not None:                           # 1x1: OIHW and OHWI are the same order
                return sv.reshape(w.shape[0], w.shape[1])

    def build():
        Co, Ci, kh, kw = w.shape
        cp = ops.round_up(Ci, ops.epc(dtype))
        t = w.detach().to(dtype).permute(0, 2, 3, 1)
        if cp != Ci:
            t = torch.nn.functional.pad(t, (0, cp - Ci))
        return t.reshape(Co, kh * kw * cp).contiguous()

    return cached(conv, "packed_%s" % dtype, (w,), build)


def packed_conv_weight_dgrad(conv, dtype):
    """w_dgrad[ci][(kh-1-ky, kw-1-kx, co)] with ci / co padded to 16-byte multiples (include/e2eft.h e2eft_conv2d_dgrad)."""
    w = conv.weight

    def build():
        Co, Ci, kh, kw = w.shape
        e = ops.epc(dtype)
        cip, cop = ops.round_up(Ci, e), ops.round_up(Co, e)
        if w.is_cuda:      # one batched transpose of the packed forward weight (for FlatAdamW-resident weights that is a view of the 16-bit twin: no cast, no flip, no strided copy)
            pk = packed_conv_weight(conv, dtype)
            if pk.is_contiguous() and pk.data_ptr() % 16 == 0:
                return ops.dgrad_weight_from_packed(pk, Co, kh * kw, cip)
        t = w.detach().to(dtype).flip(2, 3).permute(1, 2, 3, 0)        # [Ci, kh, kw, Co]
        t = torch.nn.functional.pad(t, (0, cop - Co, 0, 0, 0, 0, 0, cip - Ci))
        return t.reshape(cip, kh * kw * cop).contiguous()

    return cached(conv, "packed_dgrad_%s" % dtype, (w,), build)


UPCONV_DGRAD_4X4 = True     # tests / A-B: False sends the data gradient of upsample-convolutions through the 3x3 dgrad on the upsampled grid + e2eft_upsample_nearest_bwd


def upconv_dgrad_weight(conv, dtype):
    """The data gradient of `nearest-2x upsample -> conv 3x3 / pad 1` w.r.t. the LOW-resolution input is ONE 4x4 / stride-2 / pad-1 convolution of dY:
        dX[Y, X, ci] = sum_{u, v in 0..3, co} wt[ci][u][v][co] * dY[2Y - 1 + u, 2X - 1 + v, co],   wt[ci][u][v][co] = sum_{ky in S(u), kx in S(v)} w[co][ci][ky][kx],
    S(0) = {2}, S(1) = {1, 2}, S(2) = {0, 1}, S(3) = {0}: source pixel (Y, X) was read by the output pixels (2Y + a - ky + 1, ...), a in {0, 1} — the transpose of
    autograd.phase_conv_weight's four 2x2 phases.  16 / 36 = 4/9 of the multiply-adds of the 3x3 dgrad on the upsampled grid, and the fold-back pass
    (e2eft_upsample_nearest_bwd) disappears.  -> [Ci_pad, 4 * 4 * Co_pad] rows (u, v, co), summed in fp32 from the master weights, cached per parameter version."""
    w = conv.weight

    def build():
        Co, Ci = w.shape[:2]
        e = ops.epc(dtype)
        cip, cop = ops.round_up(Ci, e), ops.round_up(Co, e)
        w32 = w.detach().to(torch.float64 if w.dtype == torch.float64 else torch.float32)      # [Co, Ci, 3, 3]
        S = ((2,), (1, 2), (0, 1), (0,))
        out = torch.zeros((cip, 4, 4, cop), dtype=w32.dtype, device=w32.device)
        for u in range(4):
            rows = sum(w32[:, :, ky, :] for ky in S[u])                 # [Co, Ci, 3]
            for v in range(4):
                out[:Ci, u, v, :Co] = sum(rows[:, :, kx] for kx in S[v]).t()
        return out.reshape(cip, 16 * cop).to(dtype).contiguous()

    return cached(conv, "upconv_dgrad_%s" % dtype, (w,), build)


def _dense_nhwc(g, cpad):
    """gradient tensor [B,H,W,C] -> pixel-dense NHWC view/copy whose channel extent is padded to `cpad` with zeros"""
    B, H, W, Cc = g.shape
    ok = True
    try:
        ld = ops._nhwc_ld(g)
        ok = ld % ops.epc(g.dtype) == 0 and g.data_ptr() % 16 == 0
    except ValueError:
        ok = False
    if ok and Cc == cpad:
        return g
    out = torch.zeros((B, H, W, cpad), dtype=g.dtype, device=g.device) if cpad != Cc else torch.empty((B, H, W, cpad), dtype=g.dtype, device=g.device)
    if ok:
        ops.copy_scale(g, out[..., :Cc])
    else:
        out[..., :Cc].copy_(g)
    return out


def _rows(t):
    """[..., C] tensor -> 2-D row view [M, C] (copy only when the leading dims are not dense)"""
    Cc = t.shape[-1]
    if t.is_contiguous():
        return t.reshape(-1, Cc)
    try:
        v = ops._as_rows(t)
        if v.stride(0) % ops.epc(t.dtype) == 0 and v.data_ptr() % 16 == 0:
            return v
    except ValueError:
        pass
    return t.contiguous().reshape(-1, Cc)


# GroupNorm statistics emitted by a producer's epilogue ride on the output tensor as a Python attribute (ops.GnStats).  An
# autograd Function returns a fresh tensor object, so forward() parks the statistics here and the wrapper re-attaches them.
_LAST_STATS = [None]


def _stash_stats(out):
    _LAST_STATS[0] = getattr(out, "_e2eft_gn", None)


def _attach_stats(y):
    st, _LAST_STATS[0] = _LAST_STATS[0], None
    if st is not None:
        y._e2eft_gn = st
    return y


# ------------------------------------------------------------------------------------------------------------
class _Conv2dFn(torch.autograd.Function):
    """out = alpha * (conv(cat(x, x2)) + bias + rowadd[b]) + residual — forward e2eft_conv2d_fwd, backward e2eft_conv2d_dgrad
    (+ e2eft_upsample_nearest_bwd), e2eft_transpose + e2eft_conv2d_im2col_t + e2eft_gemm (wgrad), e2eft_colsum (bias / rowadd)."""

    @staticmethod
    def forward(ctx, x, x2, weight, bias, rowadd, residual, conv, stride, pad, up_to, alpha, gn_stats):
        dt = x.dtype
        kh, kw = weight.shape[2:]
        cout = weight.shape[0]
        xp = ops.pad_channels(x) if x2 is None else x     # 3/4-channel inputs: zero padded copy (weights are packed to match)
        out = ops.conv2d(xp, packed_conv_weight(conv, dt), _vec(bias, dt), cout, kh, kw, stride, pad, x2=x2, up_to=up_to,
                         rowadd=rowadd, residual=residual, alpha=alpha, gn_stats=gn_stats,
                         w_phase=(lambda: phase_conv_weight(conv, dt)) if (up_to is not None and (kh, kw) == (3, 3)) else None)
        _stash_stats(out)
        ctx.save_for_backward(xp, x2, weight, bias)
        ctx.conv, ctx.geom = conv, (stride, pad, up_to, alpha, x.shape[3])
        ctx.has = (rowadd is not None, residual is not None)
        return out

    @staticmethod
    def backward(ctx, dy):
        xp, x2, weight, bias = ctx.saved_tensors
        stride, pad, up_to, alpha, c_orig = ctx.geom
        conv = ctx.conv
        dt = xp.dtype
        Co, Ci, kh, kw = weight.shape
        B, H, W, c1 = xp.shape
        c2 = 0 if x2 is None else x2.shape[3]
        need = ctx.needs_input_grad
        e = ops.epc(dt)
        cop = ops.round_up(Co, e)
        dyp = _dense_nhwc(dy, cop)
        dx = dx2 = dw = dbias = drow = dres = None
        if ctx.has[1] and need[5]:
            dres = dy
        if (bias is not None and need[3]) or (ctx.has[0] and need[4]):
            bsink = grad_sink(conv.bias) if (bias is not None and need[3] and conv.bias is not None and conv.bias.dtype == torch.float32) else None
            # per-image column sums (B groups: B x the workgroups of one reduction over all rows), then the sum over the images — written into the bias
            # gradient's slot of the flat buffer when this call is its first writer (same kernels, same order either way)
            s = ops.colsum(ops._as_rows(dyp), groups=B, alpha=alpha)[:, :Co]       # [B, Co] fp32
            if ctx.has[0] and need[4]:
                drow = s.to(dt)
            if bias is not None and need[3]:
                if bsink is not None:
                    torch.sum(s, 0, out=bsink[0])
                    dbias = bsink[1][0]
                else:
                    dbias = s.sum(0).to(bias.dtype)
        if need[0] or (x2 is not None and need[1]):
            if (UPCONV_DGRAD_4X4 and up_to is not None and x2 is None and (kh, kw, stride) == (3, 3, 1) and tuple(pad) == (1, 1, 1, 1)
                    and tuple(up_to) == (2 * H, 2 * W) and tuple(dyp.shape[1:3]) == (2 * H, 2 * W)):
                # upsampler: one 4x4 / stride-2 convolution of dY lands directly on the low-resolution grid (4/9 of the multiply-adds, no fold-back pass)
                dxl = ops.conv2d(dyp, upconv_dgrad_weight(conv, dt), None, c1, 4, 4, 2, (1, 1, 1, 1), alpha=alpha)
            else:
                dxl = ops.conv2d_dgrad(dyp, packed_conv_weight_dgrad(conv, dt), (B, H, W, c1), c2, kh, kw, stride, pad, up_to, alpha)
                if up_to is not None:
                    dxl = ops.upsample_nearest_bwd(dxl, H, W)
            if need[0]:
                dx = dxl[..., :c_orig]
            if x2 is not None and need[1]:
                dx2 = dxl[..., c1:]
        if need[2]:
            # straight from the NHWC tensors (csrc/wgrad.hip: transpose reads, no dY^T / im2col copies); None -> the GEMM path below (fp32, fused
            # upsample, channel counts that are not multiples of 64)
            # FlatAdamW's slot of this weight's gradient (None unless this call is its first writer): the reduction's result lands there when the slot has the
            # kernels' layout — OHWI rows without channel padding (1x1: OIHW is the same order)
            wsink = None
            if weight.dtype == torch.float32 and c1 + c2 == Ci and ((kh == 1 and kw == 1 and weight.is_contiguous()) or _is_ohwi(weight)):
                wsink = grad_sink(conv.weight)
            dwp = ops.conv2d_wgrad(dyp, xp, x2, Co, kh, kw, stride, pad, alpha, out=None if wsink is None else wsink[0]) if up_to is None else None
            if dwp is None:
                P = dyp.shape[0] * dyp.shape[1] * dyp.shape[2]
                nsplit, kc = ops.splitk_plan(Co, kh * kw * (c1 + c2), P, dt)
                dyT = ops.transpose(ops._as_rows(dyp), rows_pad=nsplit * kc)        # [cop, Pp]
                col, _, _ = ops.im2col_t(xp, x2, kh, kw, stride, pad, up_to, Pp=nsplit * kc)   # [kh*kw*cin, Pp]
                dwp = ops.gemm_splitk(dyT[:Co], col, nsplit, kc, alpha=alpha)       # [Co, kh*kw*cin] = OHWI
            # the gradient is handed to autograd in the PARAMETER's stride order (OIHW contiguous): AccumulateGrad then adds / stores it
            # without a strided pass, and a DistributedDataParallel wrap finds "grad strides == bucket view strides"
            if wsink is not None:
                if dwp.data_ptr() != wsink[0].data_ptr():      # a fallback path produced it elsewhere: one copy instead of AccumulateGrad's add
                    wsink[0].view(dwp.shape).copy_(dwp)
                dw = wsink[1][0]
            elif kh == 1 and kw == 1:
                dw = dwp[:, :Ci].reshape(Co, Ci, 1, 1)
                dw = dw if dw.dtype == weight.dtype else dw.to(weight.dtype)
            elif _is_ohwi(weight):     # the master weight lies in the kernels' order (FlatAdamW): the gradient is the reduction's output itself, no permuting copy
                dw = dwp.view(Co, kh, kw, c1 + c2)[..., :Ci].permute(0, 3, 1, 2)
                dw = dw if dw.dtype == weight.dtype else dw.to(weight.dtype)
            else:
                dw = dwp.view(Co, kh, kw, c1 + c2)[..., :Ci].permute(0, 3, 1, 2).to(dtype=weight.dtype, memory_format=torch.contiguous_format)
        return dx, dx2, dw, dbias, drow, dres, None, None, None, None, None, None


def conv(conv_mod, x, x2=None, up_to=None, rowadd=None, residual=None, alpha=1.0, pad=None, gn_stats=True, norm=None):
    """nn.Conv2d-shaped module on NHWC input (see modules.conv_nhwc); differentiable when a gradient is required.
    norm = (GroupNorm module, silu): the convolution of norm(x) — without autograd the library may apply the norm inside the convolution's
    operand fetch (ops.conv2d); under autograd the norm runs as its own differentiable op first."""
    if norm is not None and (needs_grad(x, x2, conv_mod.weight, conv_mod.bias, rowadd, residual, norm[0].weight, norm[0].bias) or x2 is not None):
        x = groupnorm(x, norm[0].weight, norm[0].bias, norm[0].num_groups, norm[0].eps, silu=norm[1], x2=x2)
        x2, norm = None, None
    kh, kw = conv_mod.weight.shape[2:]
    stride = conv_mod.stride[0] if isinstance(conv_mod.stride, tuple) else conv_mod.stride
    if pad is None:
        p = conv_mod.padding[0] if isinstance(conv_mod.padding, tuple) else conv_mod.padding
        pad = (p, p, p, p)
    if needs_grad(x, x2, conv_mod.weight, conv_mod.bias, rowadd, residual):
        return _attach_stats(_Conv2dFn.apply(x, x2, conv_mod.weight, conv_mod.bias, rowadd, residual, conv_mod, stride, tuple(pad), up_to, alpha, gn_stats))
    dt = x.dtype
    if x2 is None:
        x = ops.pad_channels(x)
    nrm = None if norm is None else (_vec(norm[0].weight, dt), _vec(norm[0].bias, dt), norm[0].num_groups, norm[0].eps, norm[1])
    return ops.conv2d(x, packed_conv_weight(conv_mod, dt), _vec(conv_mod.bias, dt), conv_mod.weight.shape[0], kh, kw, stride, pad, x2=x2,
                      up_to=up_to, rowadd=rowadd, residual=residual, alpha=alpha, gn_stats=gn_stats, norm=nrm,
                      w_phase=(lambda: phase_conv_weight(conv_mod, dt)) if (up_to is not None and (kh, kw) == (3, 3)) else None)


# ------------------------------------------------------------------------------------------------------------
def _cat_weight(owner, name, weights, dtype, kpad):
    def build():
        w = torch.cat([p.detach().to(dtype) for p in weights], dim=0) if len(weights) > 1 else weights[0].detach().to(dtype)
        if kpad != w.shape[1]:
            w = torch.nn.functional.pad(w, (0, kpad - w.shape[1]))
        return w.contiguous()

    if len(weights) == 1 and weights[0].dtype == dtype and kpad == weights[0].shape[1]:
        return weights[0].detach()
    if kpad == weights[0].shape[1] and weights[0].dtype != dtype:
        # q | k | v (k | v) of one attention lie back to back in the flat master buffer (module order, no padding between them when N K % 4 == 0): their
        # concatenation is one contiguous slice of the twin
        tags = [getattr(p, "_e2eft_flat", None) for p in weights]
        if all(t is not None for t in tags) and all(t[0] is tags[0][0] for t in tags) and all(p.is_contiguous() and p.dim() == 2 for p in weights):
            offs = [t[1] for t in tags]
            if all(offs[i] + weights[i].numel() == offs[i + 1] for i in range(len(weights) - 1)):
                first = shadow_view(weights[0], dtype)
                if first is not None and all(shadow_view(p, dtype) is not None for p in weights[1:]):
                    rows = sum(p.shape[0] for p in weights)
                    return tags[0][0].twin(dtype).as_strided((rows, kpad), (kpad, 1), offs[0])
    return cached(owner, "%s_%s" % (name, dtype), tuple(weights), build)


def _cat_weight_t(owner, name, weights, dtype, kpad):
    """[K_pad, N_pad64] transpose of the concatenated weight: the W operand of dX = dY · W"""
    return cached(owner, "%s_t_%s" % (name, dtype), tuple(weights), lambda: ops.transpose(_cat_weight(owner, name, weights, dtype, kpad)))


class _LinearFn(torch.autograd.Function):
    """y = alpha * (x W^T + b) + residual with W = cat(weights) along the output dim (fused q/k/v projections)."""

    @staticmethod
    def forward(ctx, x, bias, residual, owner, name, alpha, gn_rows, *weights):
        dt = x.dtype
        K = x.shape[-1]
        kp = ops.round_up(K, ops.epc(dt))
        xp = torch.nn.functional.pad(x, (0, kp - K)) if kp != K else x
        w = _cat_weight(owner, name, weights, dt, kp)
        y = ops.linear(xp, w, _vec(bias, dt), residual=residual, alpha=alpha, gn_rows_per_image=gn_rows)
        _stash_stats(y)
        ctx.save_for_backward(xp, bias, *weights)
        ctx.meta = (owner, name, alpha, K, residual is not None)
        ctx.psrc = (bias, weights)      # the Parameter objects themselves (grad_sink reads their tags): under activation recompute saved_tensors hands back detached aliases
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, bias, *weights = ctx.saved_tensors
        owner, name, alpha, K, has_res = ctx.meta
        dt = xp.dtype
        kp = xp.shape[-1]
        need = ctx.needs_input_grad
        g = _rows(dy)                                    # [M, N]
        M, N = g.shape
        dx = dbias = dres = None
        dws = [None] * len(weights)
        if has_res and need[2]:
            dres = dy
        if need[0]:
            wt = _cat_weight_t(owner, name, weights, dt, kp)        # [kp, N64]
            n64 = wt.shape[1]
            if n64 != N:   # contraction length must match: zero-extend dy's columns (N is a multiple of 64 for every UNet layer)
                g64 = torch.zeros((M, n64), dtype=dt, device=g.device)
                g64[:, :N].copy_(g)
            else:
                g64 = g
            dxp = ops.gemm(g64, wt, alpha=alpha)                    # [M, kp]
            dx = dxp.view(*xp.shape[:-1], kp)[..., :K]
        if any(need[7:]):
            # q | k | v of one attention lie back to back in FlatAdamW's buffers: one reduction writes all their gradients in place (see grad_sink)
            wsink = None
            if all(need[7:]) and kp == K and all(w_.dtype == torch.float32 and w_.dim() == 2 and w_.is_contiguous() for w_ in weights):
                wsink = grad_sink(*ctx.psrc[1])
            dw = ops.linear_wgrad(g, _rows(xp), alpha, out=None if wsink is None else wsink[0]) if M >= 512 else None      # csrc/wgrad.hip (1x1 case); None -> the transposes + GEMM below
            if dw is None:
                nsplit, kc = ops.splitk_plan(N, kp, M, dt)
                gT = ops.transpose(g, rows_pad=nsplit * kc)             # [N, Mp]
                xT = ops.transpose(_rows(xp), rows_pad=nsplit * kc)     # [kp, Mp]
                dw = ops.gemm_splitk(gT, xT, nsplit, kc, alpha=alpha)   # [N, kp]
            if wsink is not None:
                if dw.data_ptr() != wsink[0].data_ptr():
                    wsink[0].view(N, K).copy_(dw)
                dws = list(wsink[1])
            else:
                o = 0
                for i, wgt in enumerate(weights):
                    n = wgt.shape[0]
                    if need[7 + i]:
                        t = dw[o:o + n, :K]
                        dws[i] = t if t.dtype == wgt.dtype else t.to(wgt.dtype)
                    o += n
        if bias is not None and need[1]:
            bsink = grad_sink(ctx.psrc[0]) if (bias.dtype == torch.float32 and bias.dim() == 1) else None
            if bsink is not None:
                ops.colsum(g, groups=1, alpha=alpha, out=bsink[0].view(1, N))
                dbias = bsink[1][0]
            else:
                dbias = ops.colsum(g, groups=1, alpha=alpha)[0].to(bias.dtype)
        return (dx, dbias, dres, None, None, None, None, *dws)


def linear(x, weights, bias=None, residual=None, alpha=1.0, owner=None, name="w", gn_rows_per_image=0):
    """nn.Linear on the last dim; `weights` is a Parameter or a tuple of Parameters concatenated along the output dim.
    `owner` (a module) holds the cache of the concatenated / cast / transposed weight."""
    if not isinstance(weights, (tuple, list)):
        weights = (weights,)
    if needs_grad(x, bias, residual, *weights):
        return _attach_stats(_LinearFn.apply(x, bias, residual, owner, name, alpha, gn_rows_per_image, *weights))
    dt = x.dtype
    K = x.shape[-1]
    kp = ops.round_up(K, ops.epc(dt))
    if kp != K:
        x = torch.nn.functional.pad(x, (0, kp - K))
    return ops.linear(x, _cat_weight(owner, name, weights, dt, kp), _vec(bias, dt), residual=residual, alpha=alpha, gn_rows_per_image=gn_rows_per_image)


# ------------------------------------------------------------------------------------------------------------
class _GroupNormFn(torch.autograd.Function):
    """GroupNorm(+SiLU).  split=True additionally returns an alias of x for the skip path of a residual block: the gradient that
    comes back over it is added inside the backward apply kernel (e2eft_groupnorm_bwd_add) instead of by a separate torch add."""

    @staticmethod
    def forward(ctx, x, x2, gamma, beta, groups, eps, silu, s1, s2, split):
        dt = x.dtype
        y, ws = ops.groupnorm_fwd_ws(x, _vec(gamma, dt), _vec(beta, dt), groups, eps, silu=silu, x2=x2, s1=s1, s2=s2)
        ctx.save_for_backward(x, x2, gamma, beta, ws)
        ctx.meta = (groups, eps, silu)
        ctx.psrc = (gamma, beta)        # (see _LinearFn)
        if split:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, x2, gamma, beta, ws = ctx.saved_tensors
        groups, eps, silu = ctx.meta
        dt = x.dtype
        need = ctx.needs_input_grad
        c1 = x.shape[3]
        Cc = c1 + (0 if x2 is None else x2.shape[3])
        want_dx = need[0] or (x2 is not None and need[1])
        want_p = need[2] or need[3]
        if dy is None:     # only the skip path was used downstream
            return dskip, None, None, None, None, None, None, None, None, None
        g = _dense_nhwc(dy, Cc)
        add = _dense_nhwc(dskip, Cc) if (dskip is not None and need[0] and x2 is None) else None
        gs = grad_sink(ctx.psrc[0]) if (need[2] and need[3] and gamma.dtype == torch.float32) else None
        bs = grad_sink(ctx.psrc[1]) if (gs is not None and beta.dtype == torch.float32) else None
        dx, dg, db = ops.groupnorm_bwd(x, x2, _vec(gamma, dt), _vec(beta, dt), groups, eps, silu, g, ws, need_dx=want_dx, need_dparams=want_p,
                                       dx_add=add, dg_out=None if gs is None else gs[0], db_out=None if bs is None else bs[0])
        d1 = dx[..., :c1] if (need[0] and dx is not None) else None
        d2 = dx[..., c1:] if (x2 is not None and need[1]) else None
        dgam = None if not need[2] else gs[1][0] if gs is not None else dg.to(gamma.dtype)
        dbet = None if not need[3] else bs[1][0] if bs is not None else db.to(beta.dtype)
        return d1, d2, dgam, dbet, None, None, None, None, None, None


def groupnorm(x, gamma, beta, groups, eps, silu=False, x2=None, split=False):
    """split=True: returns (y, x_skip) — use x_skip wherever the block's residual reads x (see _GroupNormFn)"""
    if needs_grad(x, x2, gamma, beta):
        sp = split and x2 is None and x.requires_grad
        out = _GroupNormFn.apply(x, x2, gamma, beta, groups, eps, silu, getattr(x, "_e2eft_gn", None),
                                 getattr(x2, "_e2eft_gn", None) if x2 is not None else None, sp)
        if split:
            return out if sp else (out, x)
        return out
    dt = x.dtype
    y = ops.groupnorm(x, _vec(gamma, dt), _vec(beta, dt), groups, eps, silu=silu, x2=x2)
    return (y, x) if split else y


class _NormConvSplitFn(torch.autograd.Function):
    """fp32, FROZEN 3x3 / stride-1 / pad-1 convolution of GroupNorm(+SiLU)(x) (+ residual) — the VAE decoder's ResnetBlock2D halves under the fp32 recipe
    (training/scripts/train_marigold_e2e_ft_depth.sh:15; the VAE carries no gradient: training/train.py:304).  Forward: the norm's apply pass writes the f16 split
    planes (e2eft_groupnorm_fwd_split), the convolution multiplies them on the f16 matrix pipe (e2eft_conv2d_fwd_f32split) — the normalised fp32 tensor is neither
    written nor kept.  Backward: the data gradient is the same convolution call on dY (ops.conv2d_dgrad), then e2eft_groupnorm_bwd_add.  split as _GroupNormFn."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, conv, groups, eps, silu, s1, split, gn_stats):
        dt = x.dtype
        B, H, W, c1 = x.shape
        cout = conv.weight.shape[0]
        planes, pscale, ws = ops.groupnorm_fwd_split_ws(x, _vec(gamma, dt), _vec(beta, dt), groups, eps, silu=silu, s1=s1)
        out = ops.new_nhwc(B, H, W, cout, dt, x.device)
        if residual is not None:
            assert tuple(residual.shape) == tuple(out.shape) and residual.dtype == dt
        r = ops._conv2d_f32split(None, packed_conv_weight(conv, dt), _vec(conv.bias, dt), cout, residual, 1.0, out, gn_stats and ops.GN_STATS_ENABLED and cout % 8 == 0,
                                 "conv3x3s1n B%d %dx%d %d->%d" % (B, H, W, c1, cout), planes=planes, scale=pscale)
        assert r is not None, "norm_conv_split: the library declined a shape ops.f32split_shape_ok accepted"
        _stash_stats(out)
        ctx.save_for_backward(x, gamma, beta, ws)
        ctx.conv, ctx.meta, ctx.has_res = conv, (groups, eps, silu), residual is not None
        ctx.psrc = (gamma, beta)
        if split:
            return out, x.view_as(x)
        return out

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, gamma, beta, ws = ctx.saved_tensors
        groups, eps, silu = ctx.meta
        conv = ctx.conv
        dt = x.dtype
        need = ctx.needs_input_grad
        B, H, W, c1 = x.shape
        if dy is None:
            return dskip, None, None, None, None, None, None, None, None, None, None
        Co = conv.weight.shape[0]
        cop = ops.round_up(Co, ops.epc(dt))
        dyp = _dense_nhwc(dy, cop)
        dres = dy if (ctx.has_res and need[3]) else None
        dx = dgam = dbet = None
        if need[0] or need[1] or need[2]:
            dhn = ops.conv2d_dgrad(dyp, packed_conv_weight_dgrad(conv, dt), (B, H, W, c1), 0, 3, 3, 1, (1, 1, 1, 1), None, 1.0)
            add = _dense_nhwc(dskip, c1) if (dskip is not None and need[0]) else None
            want_p = need[1] or need[2]
            gs = grad_sink(ctx.psrc[0]) if (need[1] and need[2] and gamma.dtype == torch.float32) else None
            bs = grad_sink(ctx.psrc[1]) if (gs is not None and beta.dtype == torch.float32) else None
            dx, dg, db = ops.groupnorm_bwd(x, None, _vec(gamma, dt), _vec(beta, dt), groups, eps, silu, dhn, ws, need_dx=need[0], need_dparams=want_p,
                                           dx_add=add, dg_out=None if gs is None else gs[0], db_out=None if bs is None else bs[0])
            dgam = None if not need[1] else gs[1][0] if gs is not None else dg.to(gamma.dtype)
            dbet = None if not need[2] else bs[1][0] if bs is not None else db.to(beta.dtype)
            if not need[0]:
                dx = None
        elif dskip is not None:
            dx = dskip
        return dx, dgam, dbet, dres, None, None, None, None, None, None, None


def norm_conv_split(conv_mod, norm_mod, x, silu, residual=None, split=False, gn_stats=True):
    """conv_mod(SiLU?(norm_mod(x))) (+ residual) as _NormConvSplitFn, or None when that route does not apply (the caller then runs the two ops): fp32 on the GPU, a
    frozen 3x3 / stride-1 / pad-1 convolution the library takes from split planes, and a gradient to carry (without one ops.conv2d(norm=) fuses the same way)."""
    w = conv_mod.weight
    if x.dtype != torch.float32 or w.dtype != torch.float32 or not x.is_cuda or w.requires_grad or (conv_mod.bias is not None and conv_mod.bias.requires_grad):
        return None
    if not needs_grad(x, norm_mod.weight, norm_mod.bias, residual):
        return None
    stride = conv_mod.stride[0] if isinstance(conv_mod.stride, tuple) else conv_mod.stride
    pad = conv_mod.padding[0] if isinstance(conv_mod.padding, tuple) else conv_mod.padding
    B, H, W, c1 = x.shape
    if tuple(w.shape[2:]) != (3, 3) or stride != 1 or pad != 1 or w.shape[1] != c1 or not ops.f32split_shape_ok(B, H, W, c1, w.shape[0]):
        return None
    try:
        if ops._nhwc_ld(x) % 4 != 0 or x.data_ptr() % 16 != 0:
            return None
    except ValueError:
        return None
    sp = split and x.requires_grad
    out = _NormConvSplitFn.apply(x, norm_mod.weight, norm_mod.bias, residual, conv_mod, norm_mod.num_groups, norm_mod.eps, silu, getattr(x, "_e2eft_gn", None), sp, gn_stats)
    if split:
        return (_attach_stats(out[0]), out[1]) if sp else (_attach_stats(out), x)
    return _attach_stats(out)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        dt = x.dtype
        y = ops.layernorm(x, _vec(gamma, dt), _vec(beta, dt), eps)
        ctx.save_for_backward(x, gamma, beta)
        ctx.eps = eps
        ctx.psrc = (gamma, beta)        # (see _LinearFn)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta = ctx.saved_tensors
        need = ctx.needs_input_grad
        g = dy if dy.is_contiguous() else dy.contiguous()
        sink = grad_sink(*ctx.psrc) if (need[1] and need[2] and gamma.dtype == torch.float32 and beta.dtype == torch.float32) else None   # weight | bias: adjacent slots
        dx, dg, db = ops.layernorm_bwd(x, _vec(gamma, x.dtype), ctx.eps, g, need_dx=need[0], gb_out=None if sink is None else sink[0].view(2, -1))
        if sink is not None:
            return dx, sink[1][0], sink[1][1], None
        return dx, (dg.to(gamma.dtype) if need[1] else None), (db.to(beta.dtype) if need[2] else None), None


def layernorm(x, gamma, beta, eps):
    if needs_grad(x, gamma, beta):
        return _LayerNormFn.apply(x, gamma, beta, eps)
    return ops.layernorm(x, _vec(gamma, x.dtype), _vec(beta, x.dtype), eps)


class _GegluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        ctx.save_for_backward(h)
        return ops.geglu(h)

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        return ops.geglu_bwd(h, dy if dy.stride(-1) == 1 else dy.contiguous())


def geglu(h):
    return _GegluFn.apply(h) if needs_grad(h) else ops.geglu(h)


class _SiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.silu(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.silu_bwd(x, dy)


def silu(x):
    return _SiluFn.apply(x) if needs_grad(x) else ops.silu(x)


def add(a, b):
    """tiny [B, temb] additions of the embedding path: torch's own op records the graph when one is needed"""
    return a + b if needs_grad(a, b) else ops.add(a, b)


# ------------------------------------------------------------------------------------------------------------
def _attn_views(qkv, kv, C):
    if kv is None:
        return qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:3 * C]
    return qkv, kv[..., :C], kv[..., C:2 * C]


def _scores(q, k, heads, dt):
    """S[b,h] = q_h k_h^T as [B, heads, N, nkp] (pad columns unwritten)"""
    B, N, C = q.shape
    Nk = k.shape[1]
    d = C // heads
    nkp = ops.round_up(Nk, ops.epc(dt))
    s = torch.empty((B, heads, N, nkp), dtype=dt, device=q.device)
    ops.bgemm_raw(dt, N, Nk, d, q, q.stride(1), (q.stride(0), d), k, k.stride(1), (k.stride(0), d), s, nkp, (heads * N * nkp, N * nkp), B, heads)
    return s, nkp


def _attn_core_unfused(q, k, v, heads, scale, causal=False):
    """softmax(q k^T * scale) v through batched MFMA GEMMs (any head dim, exact fp32 capable). q [B,N,C], k/v [B,Nk,C] views.
    causal: query i attends to keys 0..i (self-attention only; CLIP text tower)."""
    B, N, C = q.shape
    Nk = k.shape[1]
    d = C // heads
    dt = q.dtype
    s, nkp = _scores(q, k, heads, dt)
    assert not causal or N == Nk
    ops.softmax_rows_(s.view(-1, nkp), Nk, scale, causal_nq=N if causal else 0)
    vt = ops.transpose(v, rows_pad=nkp)                                         # [B, C, nkp]
    o = torch.empty((B, N, C), dtype=dt, device=q.device)
    ops.bgemm_raw(dt, N, d, nkp, s, nkp, (heads * N * nkp, N * nkp), vt, nkp, (C * nkp, d * nkp), o, C, (N * C, d), B, heads)
    return o


FUSED_FP32_ATTENTION = True   # tests / A-B: False sends fp32 attention through the GEMM + softmax form (rounds 1-4) instead of csrc/attn32.hip


def fused_attention_ok(dtype, head_dim):
    """the fused kernels serve head dim 64: fp16 / bf16 (attn.hip, attn_bwd.hip) and, since round 5, strict fp32 (attn32.hip)"""
    return head_dim == 64 and (dtype != torch.float32 or FUSED_FP32_ATTENTION)


FLASH_BACKWARD = True   # tests flip this to cross-check the fused backward against the GEMM + softmax form


class _AttentionFn(torch.autograd.Function):
    """Attention core on packed projections: self-attention takes qkv [B,N,3C]; cross-attention q [B,N,C] + kv [B,L,2C].
    Forward: the flash kernel (d = 64: 16-bit attn.hip, strict fp32 attn32.hip) or the GEMM + softmax form.  Backward: the fused kernels of csrc/attn_bwd.hip /
    attn32.hip for the flash case; otherwise (VAE's 512-dim head) it recomputes P = softmax(q k^T) and runs
        dP = dO V^T,  dS = P o (dP - rowsum(dP o P)) * scale,  dQ = dS K,  dK = dS^T Q,  dV = P^T dO
    as batched MFMA GEMMs; the pixel-major operands (K^T, Q^T, dO^T, P^T, dS^T) come from e2eft_transpose."""

    @staticmethod
    def forward(ctx, qkv, kv, heads, scale):
        C = qkv.shape[-1] // 3 if kv is None else qkv.shape[-1]
        q, k, v = _attn_views(qkv, kv, C)
        if fused_attention_ok(qkv.dtype, C // heads):
            o, lse = ops.attention(q, k, v, heads, scale, return_lse=True)
            ctx.save_for_backward(qkv, kv, o, lse)
        else:
            o = _attn_core_unfused(q, k, v, heads, scale)
            ctx.save_for_backward(qkv, kv)
        ctx.meta = (heads, scale, C)
        return o

    @staticmethod
    def backward(ctx, do):
        saved = ctx.saved_tensors          # unpacked ONCE (activation recompute re-materialises saved tensors on every access)
        qkv, kv = saved[:2]
        heads, scale, C = ctx.meta
        q, k, v = _attn_views(qkv, kv, C)
        B, N, _ = q.shape
        Nk = k.shape[1]
        d = C // heads
        dt = q.dtype
        do = do.contiguous()
        if len(saved) == 4 and FLASH_BACKWARD:      # fused backward (e2eft_attn_bwd): no N x Nk matrix in HBM
            o, lse = saved[2:]
            if kv is None:
                dqkv = torch.empty(qkv.shape, dtype=dt, device=qkv.device)
                dq, dk, dv = dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:]
                dkv = None
            else:
                dqkv = torch.empty(q.shape, dtype=dt, device=q.device)
                dkv = torch.empty(kv.shape, dtype=dt, device=q.device)
                dq, dk, dv = dqkv, dkv[..., :C], dkv[..., C:]
            ops.attention_bwd(q, k, v, o, do, lse, heads, scale, dq, dk, dv)
            return dqkv, dkv, None, None
        z = B * heads
        # P = softmax(scale * q k^T)
        p, nkp = _scores(q, k, heads, dt)
        ops.softmax_rows_(p.view(-1, nkp), Nk, scale)
        # dP = dO V^T   (contraction over d, contiguous in both)
        dp = torch.empty_like(p)
        ops.bgemm_raw(dt, N, Nk, d, do, C, (N * C, d), v, v.stride(1), (v.stride(0), d), dp, nkp, (heads * N * nkp, N * nkp), B, heads)
        ops.softmax_bwd_rows_(p.view(-1, nkp), dp.view(-1, nkp), Nk, scale)      # dp <- dS (pad columns zero)
        if kv is None:
            dqkv = torch.empty_like(qkv) if qkv.is_contiguous() else torch.empty(qkv.shape, dtype=dt, device=qkv.device)
            dq, dk, dv = dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:]
            dkv = None
        else:
            dqkv = torch.empty(q.shape, dtype=dt, device=q.device)
            dkv = torch.empty(kv.shape, dtype=dt, device=q.device)
            dq, dk, dv = dqkv, dkv[..., :C], dkv[..., C:]
        # dQ = dS K: contraction over keys -> K^T [B, C, nkp]
        kt = ops.transpose(k, rows_pad=nkp)
        ops.bgemm_raw(dt, N, d, nkp, dp, nkp, (heads * N * nkp, N * nkp), kt, nkp, (C * nkp, d * nkp), dq, dq.stride(1), (dq.stride(0), d), B, heads)
        # dK = dS^T Q, dV = P^T dO: contraction over queries -> transposed score matrices [z, nkp, Np] and Q^T / dO^T [B, C, Np]
        np_ = ops.round_up(N, 64)
        qt = ops.transpose(q, rows_pad=np_)
        dot = ops.transpose(do, rows_pad=np_)
        st = ops.transpose(dp.view(z, N, nkp), rows_pad=np_)                     # dS^T
        ops.bgemm_raw(dt, Nk, d, np_, st, np_, (heads * nkp * np_, nkp * np_), qt, np_, (C * np_, d * np_), dk, dk.stride(1), (dk.stride(0), d), B, heads)
        ops.transpose(p.view(z, N, nkp), rows_pad=np_, out=st)                   # P^T (reuses the buffer)
        ops.bgemm_raw(dt, Nk, d, np_, st, np_, (heads * nkp * np_, nkp * np_), dot, np_, (C * np_, d * np_), dv, dv.stride(1), (dv.stride(0), d), B, heads)
        return dqkv, dkv, None, None


def attention(qkv, kv, heads, scale):
    """qkv [B,N,3C] (self) or q [B,N,C] with kv [B,L,2C] (cross) -> [B,N,C]"""
    if needs_grad(qkv, kv):
        return _AttentionFn.apply(qkv, kv, heads, scale)
    C = qkv.shape[-1] // 3 if kv is None else qkv.shape[-1]
    q, k, v = _attn_views(qkv, kv, C)
    if fused_attention_ok(qkv.dtype, C // heads):
        return ops.attention(q, k, v, heads, scale)
    return _attn_core_unfused(q, k, v, heads, scale)


# ------------------------------------------------------------------------------------------------------------
class _NchwToNhwcFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cpad):
        ctx.c = x.shape[1]
        return ops.nchw_to_nhwc(x.contiguous(), cpad=cpad)

    @staticmethod
    def backward(ctx, dy):
        return ops.nhwc_to_nchw(_dense_nhwc(dy, dy.shape[3])[..., :ctx.c]), None


def nchw_to_nhwc(x, cpad):
    return _NchwToNhwcFn.apply(x, cpad) if needs_grad(x) else ops.nchw_to_nhwc(x.contiguous(), cpad=cpad)


class _DepthHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, to_unit):
        ctx.save_for_backward(x)
        ctx.to_unit = to_unit
        return ops.depth_head(x, to_unit)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.depth_head_bwd(x, dy, ctx.to_unit), None


def depth_head(x, to_unit=False):
    """decoder output NHWC [B,H,W,3] -> [B,1,H,W] = clamp(mean_c, -1, 1)  (train.py:531-533)"""
    return _DepthHeadFn.apply(x, to_unit) if needs_grad(x) else ops.depth_head(x, to_unit)


class _NormalHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, clamp, sign):
        ctx.save_for_backward(x)
        ctx.meta = (clamp, sign)
        return ops.normal_head(x, clamp=clamp, sign=sign)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.normal_head_bwd(x, dy, ctx.meta[0], ctx.meta[1]), None, None


def normal_head(x, clamp=True, sign=1.0):
    """decoder output NHWC [B,H,W,3] -> [B,3,H,W] = clamp(x / (|x| + 1e-5), -1, 1)  (train.py:535-538)"""
    return _NormalHeadFn.apply(x, clamp, sign) if needs_grad(x) else ops.normal_head(x, clamp=clamp, sign=sign)


class _SsiLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, mask):
        B = pred.shape[0]
        p = pred.reshape(B, -1).float().contiguous()
        t = target.reshape(B, -1).float().contiguous()
        m = mask.reshape(B, -1).to(torch.uint8).contiguous()
        out, ss, ws = ops.ssi_loss_fwd_saved(p, t, m)
        ctx.save_for_backward(p, t, m, ss, ws)
        ctx.shape, ctx.dtype = pred.shape, pred.dtype
        return out[0]

    @staticmethod
    def backward(ctx, g):
        p, t, m, ss, ws = ctx.saved_tensors
        dp = ops.ssi_loss_bwd(p, t, m, ss, ws, g)
        return dp.view(ctx.shape).to(ctx.dtype), None, None


def ssi_loss(pred, target, mask):
    """ScaleAndShiftInvariantLoss (training/util/loss.py:13-47) with its gradient"""
    return _SsiLossFn.apply(pred, target, mask) if needs_grad(pred) else ops.ssi_loss(pred, target, mask)


class _AngularLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, mask):
        B = pred.shape[0]
        p = pred.reshape(B, 3, -1).float().contiguous()
        t = target.reshape(B, 3, -1).float().contiguous()
        m = mask[:, 0].reshape(B, -1).to(torch.uint8).contiguous()
        out, ws = ops.angular_loss_fwd_saved(p, t, m)
        ctx.save_for_backward(p, t, m, ws)
        ctx.shape, ctx.dtype = pred.shape, pred.dtype
        return out[0]

    @staticmethod
    def backward(ctx, g):
        p, t, m, ws = ctx.saved_tensors
        dp = ops.angular_loss_bwd(p, t, m, ws, g)
        return dp.view(ctx.shape).to(ctx.dtype), None, None


def angular_loss(pred, target, mask):
    """AngularLoss (training/util/loss.py:51-67) with its gradient"""
    return _AngularLossFn.apply(pred, target, mask) if needs_grad(pred) else ops.angular_loss(pred, target, mask)
