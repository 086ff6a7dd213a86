"""Thin Python wrappers over the libe2eft C ABI (include/e2eft.h).

Tensors are torch CUDA(HIP) tensors used purely as device memory: NHWC activations are `[B, H, W, C]` tensors whose
last dim is contiguous and whose pixel stride `ld = t.stride(-2)` may exceed C (channel slice of a wider buffer).
Every function enqueues on torch's current stream and raises RuntimeError on any library error; there is no
fallback path."""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, GemmDesc, GroupNormDesc, AttnDesc, check, dtype_id


class KernelTimer:
    """Optional per-launch timing with HIP events on the launch stream (torch's current stream): bench.py uses it to
    compute achieved TFLOP/s / GB/s per kernel family over the timed region.  Inactive (zero overhead) by default."""

    def __init__(self):
        self.rec = {}

    def add(self, name, start, end, flops, nbytes, label=None):
        self.rec.setdefault(name, []).append((start, end, flops, nbytes, label))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, lst in self.rec.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in lst)
            out[name] = dict(launches=len(lst), ms=ms, flops=float(sum(r[2] for r in lst)), bytes=float(sum(r[3] for r in lst)))
        return out

    def by_label(self):
        """{(name, label): dict(launches, ms, flops, bytes)} for shape-level analysis"""
        torch.cuda.synchronize()
        out = {}
        for name, lst in self.rec.items():
            for r in lst:
                d = out.setdefault((name, r[4]), dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
                d["launches"] += 1
                d["ms"] += r[0].elapsed_time(r[1])
                d["flops"] += r[2]
                d["bytes"] += r[3]
        return out


TIMER = None  # set to a KernelTimer to record


class _timed:
    def __init__(self, name, flops=0.0, nbytes=0.0, label=None):
        self.name, self.flops, self.nbytes, self.label = name, flops, nbytes, label

    def __enter__(self):
        if TIMER is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if TIMER is not None:
            self.e.record()
            TIMER.add(self.name, self.s, self.e, self.flops, self.nbytes, self.label)
        return False


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def epc(dtype):
    """elements per 16-byte chunk"""
    return 4 if dtype == torch.float32 else 8


def round_up(v, m):
    return (v + m - 1) // m * m


def _check_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("libe2eft ops need device (HIP) tensors; got a %s tensor. There is no CPU fallback." % t.device)


def _nhwc_ld(t):
    """pixel stride of an NHWC view [B,H,W,C]; validates that pixels are densely strided"""
    assert t.dim() == 4, t.shape
    B, H, W, Cc = t.shape
    ld = t.stride(2) if W > 1 else (t.stride(1) if H > 1 else (t.stride(0) if B > 1 else max(Cc, 1)))
    if Cc > 1 and t.stride(3) != 1:
        raise ValueError("NHWC view must have unit channel stride, got strides %s" % (t.stride(),))
    if (W > 1 and t.stride(2) != ld) or (H > 1 and t.stride(1) != W * ld) or (B > 1 and t.stride(0) != H * W * ld):
        raise ValueError("NHWC view must be pixel-dense: shape %s strides %s" % (tuple(t.shape), t.stride()))
    return ld


def _rows_ld(t):
    """row stride of a 2-D row-major view [rows, C]"""
    assert t.dim() == 2, t.shape
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise ValueError("row view must have unit column stride")
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


def new_nhwc(B, H, W, Cc, dtype, device):
    """Output buffer; channel count padded to a 16-byte multiple with zeros in the pad (returned view has C channels)."""
    cp = round_up(Cc, epc(dtype))
    if cp == Cc:
        return torch.empty((B, H, W, Cc), dtype=dtype, device=device)
    return torch.zeros((B, H, W, cp), dtype=dtype, device=device)[..., :Cc]


def pad_channels(x):
    """Copy an NHWC view whose C is not a 16-byte multiple into a zero-padded buffer (rgb / 4-channel latents only)."""
    B, H, W, Cc = x.shape
    e = epc(x.dtype)
    if Cc % e == 0:
        return x
    out = torch.zeros((B, H, W, round_up(Cc, e)), dtype=x.dtype, device=x.device)
    copy_scale(x, out[..., :Cc])
    return out  # full padded view: consumers index weights packed with the same padding


class GnStats:
    """GroupNorm partial statistics emitted by the producer of a tensor (e2eft_*_gnstats): [images, nslabs, C, 3] fp32."""
    __slots__ = ("partial", "nslabs")

    def __init__(self, partial, nslabs):
        self.partial, self.nslabs = partial, nslabs


GN_STATS_ENABLED = True   # tests flip this to cross-check the fused statistics against the stand-alone pass


def _gn_buffer(images, rows_per_image, cout, device):
    nbytes = images * ((rows_per_image + 127) // 128) * cout * 12
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device), nbytes


def _attach_stats(out, buf, slab_rows, images, rows_per_image, cout):
    if slab_rows > 0:
        nslabs = rows_per_image // slab_rows
        out._e2eft_gn = GnStats(buf[: images * nslabs * cout * 3], nslabs)
    return out


# ---------------------------------------------------------------------------------------------------------
def conv2d(x, w_packed, bias, cout, kh, kw, stride=1, pad=(0, 0, 0, 0), x2=None, up_to=None, rowadd=None,
           residual=None, alpha=1.0, out=None, gn_stats=False):
    """Implicit-GEMM convolution. x: [B,H,W,C1] (C1 % epc == 0), x2: optional [B,H,W,C2] fused channel concat,
    w_packed: [cout, ldw] rows = (ky,kx,c) K-contiguous, pad = (top, bottom, left, right), up_to=(hl,wl) fused
    nearest upsample, rowadd: [B,cout] per-image vector added before alpha, residual: [B,hout,wout,cout]."""
    _check_cuda(x, w_packed, x2, bias, rowadd, residual, out)
    B, H, W, c1 = x.shape
    hl, wl = (H, W) if up_to is None else up_to
    pt, pb, pl, pr = pad
    hout = (hl + pt + pb - kh) // stride + 1
    wout = (wl + pl + pr - kw) // stride + 1
    if out is None:
        out = new_nhwc(B, hout, wout, cout, x.dtype, x.device)
    assert tuple(out.shape) == (B, hout, wout, cout), (out.shape, (B, hout, wout, cout))
    d = ConvDesc()
    d.dtype = dtype_id(x.dtype)
    d.batch, d.hin, d.win, d.hl, d.wl = B, H, W, hl, wl
    d.c1, d.ldx1 = c1, _nhwc_ld(x)
    if x2 is not None:
        assert x2.shape[:3] == x.shape[:3] and x2.dtype == x.dtype
        d.c2, d.ldx2 = x2.shape[3], _nhwc_ld(x2)
    else:
        d.c2, d.ldx2 = 0, 0
    d.kh, d.kw, d.stride, d.pad_t, d.pad_l = kh, kw, stride, pt, pl
    d.hout, d.wout, d.cout, d.ldo = hout, wout, cout, _nhwc_ld(out)
    d.ldr = _nhwc_ld(residual) if residual is not None else 0
    if residual is not None:
        assert tuple(residual.shape) == tuple(out.shape) and residual.dtype == x.dtype
    assert w_packed.dim() == 2 and w_packed.shape[0] == cout and w_packed.is_contiguous() and w_packed.dtype == x.dtype
    assert w_packed.shape[1] >= kh * kw * (d.c1 + d.c2), (w_packed.shape, kh, kw, d.c1, d.c2)
    d.ldw = w_packed.shape[1]
    d.alpha = alpha
    if bias is not None:
        assert bias.dtype == x.dtype and bias.numel() == cout and bias.is_contiguous()
    if rowadd is not None:
        assert rowadd.dtype == x.dtype and tuple(rowadd.shape) == (B, cout) and rowadd.is_contiguous()
    want = gn_stats and GN_STATS_ENABLED and cout % 8 == 0
    with _timed("igemm", 2.0 * B * hout * wout * cout * kh * kw * (d.c1 + d.c2),
                label="conv%dx%ds%d%s B%d %dx%d %d->%d" % (kh, kw, stride, "u" if up_to else "", B, hout, wout, d.c1 + d.c2, cout)):
        if want:
            buf, nbytes = _gn_buffer(B, hout * wout, cout, x.device)
            slab = C.c_int32(0)
            check(_lib.load().e2eft_conv2d_fwd_gnstats(C.byref(d), _ptr(x), _ptr(x2), _ptr(w_packed), _ptr(bias), _ptr(rowadd),
                                                       _ptr(residual), _ptr(out), _ptr(buf), nbytes, C.byref(slab), _stream()))
            _attach_stats(out, buf, slab.value, B, hout * wout, cout)
        else:
            check(_lib.load().e2eft_conv2d_fwd(C.byref(d), _ptr(x), _ptr(x2), _ptr(w_packed), _ptr(bias), _ptr(rowadd),
                                               _ptr(residual), _ptr(out), _stream()))
    return out


def gemm(a, w, bias=None, residual=None, out=None, alpha=1.0, bias_along_m=False, gn_rows_per_image=0):
    """out[m,n] = alpha*(sum_k a[m,k] w[n,k] + bias) + residual.  a: [M,K] view, w: [N,K] view.
    gn_rows_per_image > 0: also emit GroupNorm partial statistics of `out` (attached as out._e2eft_gn)."""
    _check_cuda(a, w, bias, residual, out)
    M, K = a.shape
    N, K2 = w.shape
    assert K == K2 and a.dtype == w.dtype
    if out is None:
        out = torch.empty((M, N), dtype=a.dtype, device=a.device)
    assert tuple(out.shape) == (M, N) and out.dtype == a.dtype
    d = GemmDesc()
    d.dtype = dtype_id(a.dtype)
    d.m, d.n, d.k = M, N, K
    d.lda, d.ldw, d.ldo = _rows_ld(a), _rows_ld(w), _rows_ld(out)
    d.ldr = _rows_ld(residual) if residual is not None else 0
    d.nzo = d.nzi = 1
    d.bias_along_m = 1 if bias_along_m else 0
    d.alpha = alpha
    if bias is not None:
        assert bias.dtype == a.dtype and bias.is_contiguous() and bias.numel() == (M if bias_along_m else N)
    if residual is not None:
        assert tuple(residual.shape) == (M, N) and residual.dtype == a.dtype
    want = gn_rows_per_image > 0 and GN_STATS_ENABLED and N % 8 == 0 and M % gn_rows_per_image == 0 and not bias_along_m
    with _timed("igemm", 2.0 * M * N * K, label="gemm M%d N%d K%d" % (M, N, K)):
        if want:
            buf, nbytes = _gn_buffer(M // gn_rows_per_image, gn_rows_per_image, N, a.device)
            slab = C.c_int32(0)
            check(_lib.load().e2eft_gemm_gnstats(C.byref(d), _ptr(a), _ptr(w), _ptr(bias), _ptr(residual), _ptr(out), gn_rows_per_image,
                                                 _ptr(buf), nbytes, C.byref(slab), _stream()))
            _attach_stats(out, buf, slab.value, M // gn_rows_per_image, gn_rows_per_image, N)
        else:
            check(_lib.load().e2eft_gemm(C.byref(d), _ptr(a), _ptr(w), _ptr(bias), _ptr(residual), _ptr(out), _stream()))
    return out


def bgemm_raw(dtype, m, n, k, a, lda, sa, w, ldw, sw, out, ldo, so, nzo, nzi, bias=None, bias_along_m=False, alpha=1.0):
    """Batched NT GEMM on raw base tensors with explicit element strides. sa/sw/so = (outer, inner) batch strides.
    a / w / out are tensors whose data_ptr() is the z = 0 base."""
    _check_cuda(a, w, out, bias)
    d = GemmDesc()
    d.dtype = dtype_id(dtype)
    d.m, d.n, d.k = m, n, k
    d.lda, d.ldw, d.ldo, d.ldr = lda, ldw, ldo, 0
    d.nzo, d.nzi = nzo, nzi
    d.sa_o, d.sa_i = sa
    d.sw_o, d.sw_i = sw
    d.so_o, d.so_i = so
    d.sr_o = d.sr_i = 0
    d.bias_along_m = 1 if bias_along_m else 0
    d.alpha = alpha
    with _timed("igemm", 2.0 * m * n * k * nzo * nzi, label="bgemm z%d M%d N%d K%d" % (nzo * nzi, m, n, k)):
        check(_lib.load().e2eft_gemm(C.byref(d), _ptr(a), _ptr(w), _ptr(bias), _ptr(None), _ptr(out), _stream()))
    return out


def linear(x, weight, bias=None, residual=None, out=None, alpha=1.0, gn_rows_per_image=0):
    """nn.Linear on the last dim of x [..., K] (dense rows); weight [N, K]."""
    K = x.shape[-1]
    a = x.reshape(-1, K) if x.is_contiguous() else _as_rows(x)
    r = None
    if residual is not None:
        r = residual.reshape(-1, residual.shape[-1]) if residual.is_contiguous() else _as_rows(residual)
    N = weight.shape[0]
    o2 = None
    if out is not None:
        o2 = out.reshape(-1, N) if out.is_contiguous() else _as_rows(out)
    y = gemm(a, weight, bias, r, o2, alpha, gn_rows_per_image=gn_rows_per_image)
    res = y.reshape(*x.shape[:-1], N) if out is None else out
    st = getattr(y, "_e2eft_gn", None)
    if st is not None:
        res._e2eft_gn = st
    return res


def _as_rows(t):
    """[..., C] view with dense leading dims and a row stride -> 2-D [rows, C] as_strided view."""
    Cc = t.shape[-1]
    ld = t.stride(-2)
    rows = 1
    for s in t.shape[:-1]:
        rows *= s
    # verify leading dims are dense w.r.t. ld
    exp = ld
    for i in range(t.dim() - 2, -1, -1):
        if t.shape[i] > 1 and t.stride(i) != exp:
            raise ValueError("cannot flatten view with shape %s strides %s" % (tuple(t.shape), t.stride()))
        exp *= t.shape[i]
    return t.as_strided((rows, Cc), (ld, 1))


# ---------------------------------------------------------------------------------------------------------
def groupnorm(x, gamma, beta, groups, eps, silu=False, x2=None, out=None):
    """GroupNorm(+SiLU) over NHWC x [B,H,W,C1] (optionally concatenated with x2 along C)."""
    _check_cuda(x, gamma, beta, x2, out)
    B, H, W, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[3]
    if out is None:
        out = torch.empty((B, H, W, c1 + c2), dtype=x.dtype, device=x.device)
    d = GroupNormDesc()
    d.dtype = dtype_id(x.dtype)
    d.batch, d.hw = B, H * W
    d.c1, d.ldx1 = c1, _nhwc_ld(x)
    d.c2, d.ldx2 = (c2, _nhwc_ld(x2)) if x2 is not None else (0, 0)
    d.groups, d.ldy, d.silu, d.eps = groups, _nhwc_ld(out), 1 if silu else 0, eps
    lib = _lib.load()
    nbytes = lib.e2eft_groupnorm_workspace_bytes(C.byref(d))
    if nbytes == 0:
        raise RuntimeError("groupnorm: %s" % lib.e2eft_last_error().decode())
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    if gamma is not None:
        assert gamma.dtype == x.dtype and gamma.numel() == c1 + c2
    s1 = getattr(x, "_e2eft_gn", None) if GN_STATS_ENABLED else None
    s2 = getattr(x2, "_e2eft_gn", None) if (x2 is not None and GN_STATS_ENABLED) else None
    with _timed("groupnorm", 0.0, 2.0 * B * H * W * (c1 + c2) * x.element_size(), label="gn B%d %dx%d C%d" % (B, H, W, c1 + c2)):
        check(lib.e2eft_groupnorm_fwd_pre(C.byref(d), _ptr(x), _ptr(x2), _ptr(gamma), _ptr(beta), _ptr(out),
                                          _ptr(s1.partial) if s1 else C.c_void_p(0), s1.nslabs if s1 else 0,
                                          _ptr(s2.partial) if s2 else C.c_void_p(0), s2.nslabs if s2 else 0, _ptr(ws), nbytes, _stream()))
    return out


def layernorm(x, gamma, beta, eps, out=None):
    _check_cuda(x, gamma, beta, out)
    Cc = x.shape[-1]
    a = x.reshape(-1, Cc) if x.is_contiguous() else _as_rows(x)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    o = out.reshape(-1, Cc) if out.is_contiguous() else _as_rows(out)
    assert gamma.dtype == x.dtype and beta.dtype == x.dtype
    check(_lib.load().e2eft_layernorm_fwd(dtype_id(x.dtype), a.shape[0], Cc, _rows_ld(a), _rows_ld(o), eps, _ptr(a), _ptr(gamma),
                                         _ptr(beta), _ptr(o), _stream()))
    return out


def geglu(h, out=None):
    """h [..., 2c] -> h[..., :c] * gelu(h[..., c:])"""
    _check_cuda(h, out)
    c2 = h.shape[-1]
    c = c2 // 2
    a = h.reshape(-1, c2) if h.is_contiguous() else _as_rows(h)
    if out is None:
        out = torch.empty((*h.shape[:-1], c), dtype=h.dtype, device=h.device)
    o = out.reshape(-1, c) if out.is_contiguous() else _as_rows(out)
    check(_lib.load().e2eft_geglu_fwd(dtype_id(h.dtype), a.shape[0], c, _rows_ld(a), _rows_ld(o), _ptr(a), _ptr(o), _stream()))
    return out


def softmax_rows_(s, n, scale):
    """in-place softmax(scale * s[r, :n]) over a 2-D [rows, lds] contiguous buffer; pad columns [n, roundup) zeroed."""
    _check_cuda(s)
    assert s.dim() == 2 and s.is_contiguous()
    check(_lib.load().e2eft_softmax_rows(dtype_id(s.dtype), s.shape[0], n, s.shape[1], scale, _ptr(s), _stream()))
    return s


def attention(q, k, v, heads, scale, kv_nseg=1, kv_bmod=None, out=None):
    """Fused attention, head dim 64. q: [B,Nq,heads*64] view, k/v: [Bkv,Nk,heads*64] views (row-strided)."""
    _check_cuda(q, k, v, out)
    B, Nq, Wd = q.shape
    Bkv, Nk, _ = k.shape
    assert Wd == heads * 64 and k.shape[2] == Wd and v.shape == k.shape
    if out is None:
        out = torch.empty((B, Nq, Wd), dtype=q.dtype, device=q.device)
    d = AttnDesc()
    d.dtype = dtype_id(q.dtype)
    d.batch, d.heads, d.nq, d.nk_seg = B, heads, Nq, Nk
    d.kv_nseg = kv_nseg
    d.kv_bmod = B if kv_bmod is None else kv_bmod
    assert Bkv >= d.kv_bmod * kv_nseg

    def ld3(t):
        if t.shape[2] > 1 and t.stride(2) != 1:
            raise ValueError("attention operands need unit inner stride")
        ld = t.stride(1) if t.shape[1] > 1 else max(t.shape[2], t.stride(1))
        if t.shape[0] > 1 and t.stride(0) != t.shape[1] * ld:
            raise ValueError("attention operands must be batch-dense: %s %s" % (tuple(t.shape), t.stride()))
        return ld

    d.ldq, d.ldk, d.ldv, d.ldo = ld3(q), ld3(k), ld3(v), ld3(out)
    d.scale = scale
    with _timed("attn", 4.0 * B * heads * Nq * Nk * kv_nseg * 64, label="attn B%d h%d Nq%d Nk%d" % (B, heads, Nq, Nk * kv_nseg)):
        check(_lib.load().e2eft_attn_fwd(C.byref(d), _ptr(q), _ptr(k), _ptr(v), _ptr(out), _stream()))
    return out


# ---------------------------------------------------------------------------------------------------------
def nchw_to_nhwc(x, dtype=None, cpad=None, mul=1.0, add=0.0):
    """[B,C,H,W] contiguous -> NHWC [B,H,W,cpad] (zero padded channels), y = x*mul + add."""
    _check_cuda(x)
    x = x.contiguous()
    B, Cc, H, W = x.shape
    dtype = dtype or x.dtype
    cp = round_up(Cc, epc(dtype)) if cpad is None else cpad
    y = torch.empty((B, H, W, cp), dtype=dtype, device=x.device)
    check(_lib.load().e2eft_nchw_to_nhwc(dtype_id(x.dtype), dtype_id(dtype), B, Cc, H * W, cp, cp, mul, add, _ptr(x), _ptr(y), _stream()))
    return y


def nhwc_to_nchw(x, dtype=None, mul=1.0, add=0.0):
    """NHWC view [B,H,W,C] -> contiguous [B,C,H,W]"""
    _check_cuda(x)
    B, H, W, Cc = x.shape
    dtype = dtype or x.dtype
    y = torch.empty((B, Cc, H, W), dtype=dtype, device=x.device)
    check(_lib.load().e2eft_nhwc_to_nchw(dtype_id(x.dtype), dtype_id(dtype), B, Cc, H * W, _nhwc_ld(x), mul, add, _ptr(x), _ptr(y), _stream()))
    return y


def copy_scale(x, out, mul=1.0, add=0.0):
    """out[..., :] = x * mul + add for NHWC / row views of identical shape (strided channel slices allowed)."""
    _check_cuda(x, out)
    assert x.shape == out.shape and x.dtype == out.dtype
    Cc = x.shape[-1]
    xa = _as_rows(x) if x.dim() > 2 else x
    oa = _as_rows(out) if out.dim() > 2 else out
    check(_lib.load().e2eft_copy_scale(dtype_id(x.dtype), xa.shape[0], Cc, _rows_ld(xa), _rows_ld(oa), mul, add, _ptr(xa), _ptr(oa), _stream()))
    return out


def add(a, b, out=None):
    _check_cuda(a, b, out)
    assert a.shape == b.shape and a.dtype == b.dtype
    if out is None:
        out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    Cc = a.shape[-1]
    aa, bb, oo = (_as_rows(t) if t.dim() > 2 else t for t in (a, b, out))
    check(_lib.load().e2eft_add(dtype_id(a.dtype), aa.shape[0], Cc, _rows_ld(aa), _rows_ld(bb), _rows_ld(oo), _ptr(aa), _ptr(bb), _ptr(oo), _stream()))
    return out


def timestep_embedding(t, dim, dtype):
    """t: int64 [B] device tensor -> [B, dim] = [cos | sin] (flip_sin_to_cos=True, freq_shift=0)"""
    _check_cuda(t)
    t = t.to(torch.int64).contiguous()
    out = torch.empty((t.shape[0], dim), dtype=dtype, device=t.device)
    check(_lib.load().e2eft_timestep_embedding(dtype_id(dtype), t.shape[0], dim, _ptr(t), _ptr(out), _stream()))
    return out


def silu(x):
    _check_cuda(x)
    x = x.contiguous()
    out = torch.empty_like(x)
    check(_lib.load().e2eft_silu(dtype_id(x.dtype), x.numel(), _ptr(x), _ptr(out), _stream()))
    return out


def depth_head(x, to_unit, dtype=None):
    """decoder output NHWC [B,H,W,>=3 (view)] -> [B,1,H,W] = clip(mean_c, -1, 1) (optionally mapped to [0,1])"""
    _check_cuda(x)
    B, H, W, _ = x.shape
    dtype = dtype or x.dtype
    y = torch.empty((B, 1, H, W), dtype=dtype, device=x.device)
    check(_lib.load().e2eft_depth_head(dtype_id(x.dtype), dtype_id(dtype), B * H * W, _nhwc_ld(x), 1 if to_unit else 0, _ptr(x), _ptr(y), _stream()))
    return y


def normal_head(x, clamp=False, sign=1.0, dtype=None):
    """decoder output NHWC [B,H,W,>=3] -> NCHW [B,3,H,W] unit normals n/(|n|+1e-5)"""
    _check_cuda(x)
    B, H, W, _ = x.shape
    dtype = dtype or x.dtype
    y = torch.empty((B, 3, H, W), dtype=dtype, device=x.device)
    check(_lib.load().e2eft_normal_head(dtype_id(x.dtype), dtype_id(dtype), B, H * W, _nhwc_ld(x), 1 if clamp else 0, sign, _ptr(x), _ptr(y), _stream()))
    return y


def ssi_loss(pred, target, mask, return_scale_shift=False):
    """ScaleAndShiftInvariantLoss forward (training/util/loss.py:13-47). pred/target [B,1,H,W] or [B,H,W], mask bool."""
    _check_cuda(pred, target, mask)
    B = pred.shape[0]
    p = pred.reshape(B, -1).float().contiguous()
    t = target.reshape(B, -1).float().contiguous()
    m = mask.reshape(B, -1).to(torch.uint8).contiguous()
    lib = _lib.load()
    nbytes = lib.e2eft_ssi_loss_workspace_bytes(B)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=p.device)
    out = torch.empty(1, dtype=torch.float32, device=p.device)
    ss = torch.empty((B, 2), dtype=torch.float32, device=p.device)
    check(lib.e2eft_ssi_loss_fwd(B, p.shape[1], _ptr(p), _ptr(t), _ptr(m), _ptr(out), _ptr(ss), _ptr(ws), nbytes, _stream()))
    return (out[0], ss) if return_scale_shift else out[0]


def angular_loss(pred, target, mask):
    """AngularLoss forward (training/util/loss.py:51-67). pred/target [B,3,H,W], mask [B,1,H,W] (channel 0 used)."""
    _check_cuda(pred, target, mask)
    B = pred.shape[0]
    p = pred.reshape(B, 3, -1).float().contiguous()
    t = target.reshape(B, 3, -1).float().contiguous()
    m = mask[:, 0].reshape(B, -1).to(torch.uint8).contiguous()
    lib = _lib.load()
    nbytes = lib.e2eft_angular_loss_workspace_bytes(B)
    ws = torch.empty(2, dtype=torch.float64, device=p.device)
    out = torch.empty(1, dtype=torch.float32, device=p.device)
    check(lib.e2eft_angular_loss_fwd(B, p.shape[2], _ptr(p), _ptr(t), _ptr(m), _ptr(out), _ptr(ws), nbytes, _stream()))
    return out[0]
