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

    def __init__(self, only=None):
        self.rec = {}
        self.only = only      # None: every instrumented launch; else a set of family names ("igemm", ...): the others run un-instrumented

    def add(self, name, start, end, flops, nbytes, label=None, launches=1, flops_nominal=None):
        # flops: what the kernels multiply; flops_nominal: the layer's work in the reference's formulation when that is more (e2eft_upconv2x_fwd multiplies 4/9 of it)
        self.rec.setdefault(name, []).append((start, end, flops, nbytes, label, launches, flops if flops_nominal is None else flops_nominal))      # launches: kernels of the family behind ONE library call (e2eft_upconv2x_fwd: four)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, lst in self.rec.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in lst)
            out[name] = dict(launches=sum(r[5] for r in lst), ms=ms, flops=float(sum(r[2] for r in lst)), bytes=float(sum(r[3] for r in lst)),
                             flops_nominal=float(sum(r[6] for r in lst)))
        return out

    def by_label(self):
        """{(name, label): dict(launches, ms, flops, bytes)} for shape-level analysis"""
        torch.cuda.synchronize()
        out = {}
        for name, lst in self.rec.items():
            for r in lst:
                d = out.setdefault((name, r[4]), dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
                d["launches"] += r[5]
                d["ms"] += r[0].elapsed_time(r[1])
                d["flops"] += r[2]
                d["bytes"] += r[3]
        return out


TIMER = None  # set to a KernelTimer to record


class _timed:
    def __init__(self, name, flops=0.0, nbytes=0.0, label=None, launches=1, flops_nominal=None):
        self.name, self.flops, self.nbytes, self.label, self.launches, self.flops_nominal = name, flops, nbytes, label, launches, flops_nominal

    def __enter__(self):
        self.on = TIMER is not None and (TIMER.only is None or self.name in TIMER.only)
        if self.on:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if self.on:
            self.e.record()
            label = self.label
            if self.name == "igemm":     # which kernel symbol the library's dispatch picked for this launch (debug entry, not in e2eft.h)
                label = (label, _last_kernel())
            TIMER.add(self.name, self.s, self.e, self.flops, self.nbytes, label, self.launches, self.flops_nominal)
        return False


def _last_kernel():
    lib = _lib.load()
    fn = lib.e2eft_debug_last_kernel
    fn.restype = C.c_char_p
    return fn().decode()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def dtype_size(dtype):
    return 4 if dtype == torch.float32 else 2


def epc(dtype):
    """elements per 16-byte chunk"""
    return 4 if dtype == torch.float32 else 8


def round_up(v, m):
    return (v + m - 1) // m * m


def _check_cuda(*ts):
    """Every wrapper passes its tensors through here before it launches: they must be HIP tensors of ONE device, and that device must
    be the calling thread's current device — the library launches on `torch.cuda.current_stream()`, i.e. on a stream of the CURRENT
    device, so device-1 pointers under a current device 0 would be launched on the wrong GPU.  The caller's current device is never
    changed behind their back: run a model that lives on another device inside `with torch.cuda.device(model.device):` (the pipelines'
    entry points do that themselves, `pipeline._on_device`)."""
    idx = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("libe2eft ops need device (HIP) tensors; got a %s tensor. There is no CPU fallback." % t.device)
        if idx is None:
            idx = t.device.index
        elif t.device.index != idx:
            raise RuntimeError("libe2eft ops need all tensors of a call on one device; got cuda:%d and cuda:%d" % (idx, t.device.index))
    if idx is not None and idx != torch.cuda.current_device():
        raise RuntimeError("libe2eft ops launch on the current device's stream: tensors live on cuda:%d but the current device is cuda:%d; "
                           "wrap the call in `with torch.cuda.device(%d):`" % (idx, torch.cuda.current_device(), idx))


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL = _NullCtx()


def on_device_of(t):
    """context manager: the device of tensor / module-parameter `t` is the current device inside (a no-op object when it already is)"""
    dev = t.device
    if dev.type != "cuda" or dev.index is None or dev.index == torch.cuda.current_device():
        return _NULL
    return torch.cuda.device(dev)


def device_scoped(fn):
    """decorator for the entry points of the host layer (module forwards, pipeline calls): runs the call with the device of the first
    tensor argument (or of `self`'s first parameter) as the current device and restores the caller's current device afterwards"""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        t = next((a for a in args if isinstance(a, torch.Tensor) and a.is_cuda), None)
        if t is None and isinstance(self, torch.nn.Module):
            t = next(self.parameters(), None)
        if t is None and isinstance(getattr(type(self), "device", None), property):   # pipelines: .device of their UNet
            dev = self.device
            if dev.type == "cuda" and dev.index is not None and dev.index != torch.cuda.current_device():
                with torch.cuda.device(dev):
                    return fn(self, *args, **kwargs)
            return fn(self, *args, **kwargs)
        if t is None or not t.is_cuda:
            return fn(self, *args, **kwargs)
        with on_device_of(t):
            return fn(self, *args, **kwargs)

    return wrapper


def tensor_scoped(fn):
    """the free-function twin of device_scoped (evaluate / data / ensemble entry points): the call runs with the device of its first HIP tensor argument as
    the current device — a model on cuda:1 can be evaluated from a thread whose current device is cuda:0 (these raised since _check_cuda stopped switching)"""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        t = next((a for a in list(args) + list(kwargs.values()) if isinstance(a, torch.Tensor) and a.is_cuda), None)
        if t is None:
            return fn(*args, **kwargs)
        with on_device_of(t):
            return fn(*args, **kwargs)

    return wrapper


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
SPLITK_ENABLED = True     # tests flip this to cross-check split-K convolutions against the single-pass kernel


UPCONV_PHASES_ENABLED = True     # tests / A-B: False keeps 2x-upsample convolutions on the fused-upsample 3x3 form


def _gn_buffer(images, rows_per_image, cout, device):
    nbytes = images * ((rows_per_image + 127) // 128) * cout * 12
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device), nbytes


def _attach_stats(out, buf, slab_rows, images, rows_per_image, cout):
    if slab_rows > 0:
        nslabs = rows_per_image // slab_rows
        out._e2eft_gn = GnStats(buf[: images * nslabs * cout * 3], nslabs)
    return out


F32_SPLIT_ENABLED = True      # tests / A-B: False keeps fp32 convolutions / GEMMs on the fp32 matrix instruction (igemm2)
WGRAD_F32_SPLIT = True        # tests / A-B: False keeps fp32 weight gradients on wgrad32_kernel (v_mfma_f32_32x32x2_f32)


def f32_split2(x, keep=False, x2=None):
    """fp32 NHWC [B,H,W,C] (C % 8 == 0) -> (planes f16 [B,H,W,2C] = [x0 | x1] with x * s = x0 + x1 to 2^-22, scale workspace fp32 [4]: [1] = s, [2] = 1 / s).
    s is the power of two that brings the tensor's maximum into [2^14, 2^15), found on the device (csrc/f32split.hip).
    keep: park the result on the tensor object (per version) — a gradient dY is split once for its data-gradient convolution AND its weight-gradient launches; only for
    short-lived tensors (the planes live as long as x does).  x2: a second source — the planes are then those of cat(x, x2) along the channels, under one scale."""
    _check_cuda(x, x2)
    B, H, W, Cc = x.shape
    assert x.dtype == torch.float32 and Cc % 8 == 0
    if x2 is not None:
        c2 = x2.shape[3]
        assert tuple(x2.shape[:3]) == (B, H, W) and x2.dtype == torch.float32 and c2 % 8 == 0 and not keep
        planes = torch.empty((B, H, W, 2 * (Cc + c2)), dtype=torch.float16, device=x.device)
        scale = torch.empty(4, dtype=torch.float32, device=x.device)
        with _timed("f32split", 0.0, 12.0 * B * H * W * (Cc + c2), label="split2 B%d %dx%d C%d+%d" % (B, H, W, Cc, c2), launches=4):
            check(_lib.load().e2eft_f32_split2_cat(_ptr(x), Cc, _nhwc_ld(x), _ptr(x2), c2, _nhwc_ld(x2), B * H * W, _ptr(planes), 2 * (Cc + c2), _ptr(scale), _stream()))
        return planes, scale
    if keep:
        ent = getattr(x, "_e2eft_planes", None)
        if ent is not None and ent[0] == x._version:
            return ent[1], ent[2]
    planes = torch.empty((B, H, W, 2 * Cc), dtype=torch.float16, device=x.device)
    scale = torch.empty(4, dtype=torch.float32, device=x.device)
    with _timed("f32split", 0.0, 12.0 * B * H * W * Cc, label="split2 B%d %dx%d C%d" % (B, H, W, Cc), launches=2):
        check(_lib.load().e2eft_f32_split2(_ptr(x), B * H * W, Cc, _nhwc_ld(x), _ptr(planes), 2 * Cc, _ptr(scale), _stream()))
    if keep:
        x._e2eft_planes = (x._version, planes, scale)
    return planes, scale


def f32_split_weight(w_packed, taps, c):
    """fp32 packed weight [cout, taps*c] (rows (tap, c)) -> (f16 [cout, taps*3c] rows (tap, [w0 | w1 | w0]) with w * s_w = w0 + w1, device scalar 1 / s_w).  s_w is the
    power of two that brings the weight's maximum into [2^14, 2^15), computed on the device (trainable weights change every optimizer step: no host read).  Cached on
    the packed tensor object per version (autograd.packed_conv_weight hands out one object per parameter version)."""
    ent = getattr(w_packed, "_e2eft_split", None)
    if ent is not None and ent[0] == w_packed._version:
        return ent[1], ent[2]
    rows = w_packed.numel() // c                   # cout * taps (a [4, cout, taps * c] stack of phase weights splits as one tensor: one scale)
    assert w_packed.dtype == torch.float32 and w_packed.shape[-1] == taps * c and w_packed.is_contiguous() and c % 8 == 0
    src = w_packed.detach() if w_packed.data_ptr() % 16 == 0 else w_packed.detach().clone()      # (a slice of a flat buffer may start off a 16-byte boundary)
    wsp = torch.empty((rows // taps, taps * 3 * c), dtype=torch.float16, device=w_packed.device)
    scale = torch.empty(4, dtype=torch.float32, device=w_packed.device)
    check(_lib.load().e2eft_f32_split_weight(_ptr(src), rows, c, _ptr(wsp), _ptr(scale), _stream()))      # (two launches; as tensor-library calls this was a dozen per weight)
    inv = scale[2:3]
    w_packed._e2eft_split = (w_packed._version, wsp, inv)
    return wsp, inv


def _f32split_desc(B, H, W, c1, cout, kh, kw, stride, pad, ldo, ldr):
    pt, pb, pl, pr = pad
    d = ConvDesc()
    d.dtype = _lib.F32
    d.batch, d.hin, d.win, d.hl, d.wl = B, H, W, H, W
    d.c1, d.ldx1, d.c2, d.ldx2 = c1, 2 * c1, 0, 0
    d.kh, d.kw, d.stride, d.pad_t, d.pad_l = kh, kw, stride, pt, pl
    d.hout = (H + pt + pb - kh) // stride + 1
    d.wout = (W + pl + pr - kw) // stride + 1
    d.cout, d.ldo, d.ldr, d.ldw = cout, ldo, ldr, kh * kw * 3 * c1
    d.alpha = 1.0
    return d


def f32split_shape_ok(B, H, W, c1, cout, kh=3, kw=3, stride=1, pad=(1, 1, 1, 1)):
    """pure host arithmetic: would the library run this fp32 convolution [B,H,W,c1] -> cout from f16 split planes (e2eft_conv2d_fwd_f32split_supported)?"""
    if not F32_SPLIT_ENABLED or c1 % 64 != 0 or cout % 8 != 0:
        return False
    d = _f32split_desc(B, H, W, c1, cout, kh, kw, stride, pad, cout, cout)
    return _lib.load().e2eft_conv2d_fwd_f32split_supported(C.byref(d)) == 1


def groupnorm_fwd_split_ws(x, gamma, beta, groups, eps, silu=False, s1=None):
    """fp32 GroupNorm(+SiLU) of x [B,H,W,C] whose output leaves as f16 split planes (e2eft_groupnorm_fwd_split) -> (planes [B,H,W,2C], scale workspace [4] for the
    consuming convolution, workspace for groupnorm_bwd).  The scale comes from a bound of the output, |y| <= max|gamma| * sqrt(H W C / groups) + max|beta|, computed
    on the device: no host read of the parameters, no pass over the data."""
    _check_cuda(x, gamma, beta)
    B, H, W, c1 = x.shape
    assert x.dtype == torch.float32
    planes = torch.empty((B, H, W, 2 * c1), dtype=torch.float16, device=x.device)
    scale = torch.empty(4, dtype=torch.float32, device=x.device)
    d = _gn_desc(x, None, groups, eps, silu, c1)
    lib = _lib.load()
    nbytes = lib.e2eft_groupnorm_workspace_bytes(C.byref(d))
    if nbytes == 0:
        raise RuntimeError("groupnorm: %s" % lib.e2eft_last_error().decode())
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    if not GN_STATS_ENABLED:
        s1 = None
    with _timed("groupnorm", 0.0, 2.0 * B * H * W * c1 * 4, label="gn->split B%d %dx%d C%d" % (B, H, W, c1)):
        check(lib.e2eft_groupnorm_fwd_split(C.byref(d), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(planes), 2 * c1, _ptr(scale), _ptr(s1.partial) if s1 else C.c_void_p(0),
                                            s1.nslabs if s1 else 0, _ptr(ws), nbytes, _stream()))
    return planes, scale, ws


def _conv2d_f32split(x, w_packed, bias, cout, residual, alpha, out, want, label, planes=None, scale=None, geom=(3, 3, 1, (1, 1, 1, 1)), keep_planes=False, x2=None):
    """An fp32 convolution (geom = kh, kw, stride, pads; no fused upsample) through e2eft_conv2d_fwd_f32split, or None when the library declines the
    shape.  planes / scale: the input already split (groupnorm_fwd_split_ws) — x is then ignored."""
    B, H, W, c1 = x.shape if planes is None else (planes.shape[0], planes.shape[1], planes.shape[2], planes.shape[3] // 2)
    if x2 is not None:          # two sources: the convolution of their channel concatenation (one pair of planes, one scale)
        c1 = c1 + x2.shape[3]
    kh, kw, stride, pad = geom
    d = _f32split_desc(B, H, W, c1, cout, kh, kw, stride, pad, _nhwc_ld(out), _nhwc_ld(residual) if residual is not None else 0)
    assert (d.hout, d.wout) == (out.shape[1], out.shape[2])
    H, W = d.hout, d.wout          # (below: the OUTPUT grid)
    d.alpha = alpha
    lib = _lib.load()
    if (lib.e2eft_conv2d_fwd_f32split_supported(C.byref(d)) != 1 or out.data_ptr() % 16 or (residual is not None and residual.data_ptr() % 16)
            or (bias is not None and bias.data_ptr() % 16) or not w_packed.is_contiguous() or (planes is None and (x.data_ptr() % 16 or _nhwc_ld(x) % 4))
            or (x2 is not None and (x2.data_ptr() % 16 or _nhwc_ld(x2) % 4 or x2.shape[3] % 8 or x.shape[3] % 8))):
        return None
    wsp, inv_sw = f32_split_weight(w_packed, kh * kw, c1)
    d.alpha = alpha
    if planes is None:
        planes, scale = f32_split2(x, keep=keep_planes, x2=x2)
    assert scale is not None
    nb = (planes.numel() * 2 + B * H * W * cout * 4 * (2 if residual is not None else 1) + cout * kh * kw * 3 * c1 * 2)
    # (flops: what the f16 pipe multiplies — three products per fp32 product; flops_nominal: the fp32 convolution)
    with _timed("igemm", 6.0 * B * H * W * cout * kh * kw * c1, nb, label=label + " f32split", flops_nominal=2.0 * B * H * W * cout * kh * kw * c1):
        buf, nbytes = _gn_buffer(B, H * W, cout, out.device) if want else (None, 0)
        slab = C.c_int32(0)
        check(lib.e2eft_conv2d_fwd_f32split(C.byref(d), _ptr(planes), _ptr(scale), _ptr(wsp), _ptr(inv_sw), _ptr(bias), _ptr(residual), _ptr(out), _ptr(buf), nbytes,
                                            C.byref(slab), _stream()))
        if want:
            _attach_stats(out, buf, slab.value, B, H * W, cout)
    return out


# ---------------------------------------------------------------------------------------------------------
def conv2d(x, w_packed, bias, cout, kh, kw, stride=1, pad=(0, 0, 0, 0), x2=None, up_to=None, rowadd=None,
           residual=None, alpha=1.0, out=None, gn_stats=False, norm=None, w_phase=None, _label=None):
    """Implicit-GEMM convolution. x: [B,H,W,C1] (C1 % epc == 0), x2: optional [B,H,W,C2] fused channel concat,
    w_packed: [cout, ldw] rows = (ky,kx,c) K-contiguous, pad = (top, bottom, left, right), up_to=(hl,wl) fused
    nearest upsample, rowadd: [B,cout] per-image vector added before alpha, residual: [B,hout,wout,cout].
    norm = (gamma, beta, groups, eps, silu): the convolution of GroupNorm(+SiLU)(x) — where the library can (e2eft_conv2d_fwd_normed_supported) the
    norm is never applied as a pass of its own: its statistics are finalised and the convolution reads x through them; otherwise groupnorm() runs first."""
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
    if (w_phase is not None and UPCONV_PHASES_ENABLED and F32_SPLIT_ENABLED and x.dtype == torch.float32 and up_to is not None and x2 is None and rowadd is None
            and residual is None and norm is None and alpha == 1.0 and (kh, kw, stride) == (3, 3, 1) and tuple(pad) == (1, 1, 1, 1) and (hl, wl) == (2 * H, 2 * W)
            and c1 % 64 == 0 and x.data_ptr() % 16 == 0 and d.ldx1 % 4 == 0 and out.data_ptr() % 16 == 0 and (bias is None or bias.data_ptr() % 16 == 0)):
        # fp32: the four 2x2 phases on the f16 matrix pipe from split planes (csrc/f32split.hip)
        lib = _lib.load()
        ldx_keep = d.ldx1
        d.ldx1 = 2 * c1
        ok = lib.e2eft_upconv2x_fwd_f32split_supported(C.byref(d)) == 1
        if ok:
            wp = w_phase()
            assert tuple(wp.shape) == (4, cout, 4 * c1) and wp.dtype == torch.float32
            wsp, inv_sw = f32_split_weight(wp, 4, c1)
            planes, scale = f32_split2(x)
            nbp = planes.numel() * 2 + B * hout * wout * cout * 4 + wsp.numel() * 2
            with _timed("igemm", 6.0 * B * H * W * 4 * cout * 4 * c1, nbp, label="upconv2x(4 phases) B%d %dx%d %d->%d f32split" % (B, hout, wout, c1, cout), launches=4,
                        flops_nominal=2.0 * B * hout * wout * cout * 9 * c1):
                buf, nbytes = _gn_buffer(B, hout * wout, cout, x.device) if want else (None, 0)
                slab = C.c_int32(0)
                check(lib.e2eft_upconv2x_fwd_f32split(C.byref(d), _ptr(planes), _ptr(scale), _ptr(wsp), _ptr(inv_sw), _ptr(bias), _ptr(out), _ptr(buf), nbytes,
                                                      C.byref(slab), _stream()))
                if want:
                    _attach_stats(out, buf, slab.value, B, hout * wout, cout)
            return out
        d.ldx1 = ldx_keep
    if (w_phase is not None and UPCONV_PHASES_ENABLED and up_to is not None and x2 is None and rowadd is None and residual is None and norm is None and alpha == 1.0
            and (kh, kw, stride) == (3, 3, 1) and tuple(pad) == (1, 1, 1, 1) and (hl, wl) == (2 * H, 2 * W)
            and _lib.load().e2eft_upconv2x_fwd_supported(C.byref(d)) == 1):
        wp = w_phase()
        assert tuple(wp.shape) == (4, cout, 4 * c1) and wp.is_contiguous() and wp.dtype == x.dtype, (wp.shape, wp.dtype)
        if bias is not None:
            assert bias.dtype == x.dtype and bias.numel() == cout and bias.is_contiguous()
        nbp = (B * H * W * c1 + B * hout * wout * cout + 4 * cout * 4 * c1) * x.element_size()
        with _timed("igemm", 2.0 * B * H * W * 4 * cout * 4 * c1, nbp, label="upconv2x(4 phases) B%d %dx%d %d->%d" % (B, hout, wout, c1, cout), launches=4,
                    flops_nominal=2.0 * B * hout * wout * cout * 9 * c1):
            buf, nbytes = _gn_buffer(B, hout * wout, cout, x.device) if want else (None, 0)
            slab = C.c_int32(0)
            check(_lib.load().e2eft_upconv2x_fwd(C.byref(d), _ptr(x), _ptr(wp), _ptr(bias), _ptr(out), _ptr(buf), nbytes, C.byref(slab), _stream()))
            if want:
                _attach_stats(out, buf, slab.value, B, hout * wout, cout)
        return out
    es = x.element_size()
    nb = (B * H * W * (d.c1 + d.c2) + B * hout * wout * cout * (2 if residual is not None else 1) + cout * kh * kw * (d.c1 + d.c2)) * es
    lib = _lib.load()
    sk = lib.e2eft_conv2d_splitk_workspace_bytes(C.byref(d)) if SPLITK_ENABLED else 0
    coeff = None
    if norm is not None:
        gamma, beta, groups, eps, silu = norm
        # (the <= 4-output-channel route of the supported query, narrow.hip, takes neither a row vector nor a residual nor statistics: ADVICE r3)
        narrow_ok = cout > 4 or (rowadd is None and residual is None and not want)
        if NORM_FUSION_ENABLED and x2 is None and not sk and narrow_ok and lib.e2eft_conv2d_fwd_normed_supported(C.byref(d)) == 1:
            ws, coeff = groupnorm_stats(x, gamma, groups, eps)     # (a, mean) pairs; the apply pass is the convolution's operand fetch
        elif (x.dtype == torch.float32 and x2 is None and not sk and up_to is None and rowadd is None and (kh, kw, stride) == (3, 3, 1) and tuple(pad) == (1, 1, 1, 1)
              and (alpha == 1.0 or bias is None)
              and w_packed.shape[1] == 9 * c1 and f32split_shape_ok(B, H, W, c1, cout)):
            # fp32: the norm's apply pass writes the f16 split planes the convolution reads (csrc/f32split.hip) — no fp32 intermediate, no maximum pass
            planes, pscale, _ = groupnorm_fwd_split_ws(x, gamma, beta, groups, eps, silu=silu, s1=getattr(x, "_e2eft_gn", None))
            r = _conv2d_f32split(None, w_packed, bias, cout, residual, alpha, out, want,
                                 "conv3x3s1n B%d %dx%d %d->%d" % (B, hout, wout, c1, cout), planes=planes, scale=pscale)
            if r is not None:
                return r
            x = groupnorm(x, gamma, beta, groups, eps, silu=silu)      # (the library declined after all: the two-pass route)
            d.ldx1 = _nhwc_ld(x)
        else:
            x = groupnorm(x, gamma, beta, groups, eps, silu=silu, x2=x2)      # the norm of the CONCATENATED input; its output is one tensor
            x2 = None
            d.c1, d.ldx1, d.c2, d.ldx2 = x.shape[3], _nhwc_ld(x), 0, 0
    label = _label or "conv%dx%ds%d%s%s B%d %dx%d %d->%d" % (kh, kw, stride, "u" if up_to else "", "n" if coeff is not None else "", B, hout, wout, d.c1 + d.c2, cout)
    if (F32_SPLIT_ENABLED and x.dtype == torch.float32 and coeff is None and not sk and up_to is None and rowadd is None
            and (alpha == 1.0 or bias is None)
            and (d.c1 + d.c2) % 64 == 0 and w_packed.shape[1] == kh * kw * (d.c1 + d.c2)):
        r = _conv2d_f32split(x, w_packed, bias, cout, residual, alpha, out, want, label, geom=(kh, kw, stride, tuple(pad)), x2=x2)
        if r is not None:
            return r
    with _timed("igemm", 2.0 * B * hout * wout * cout * kh * kw * (d.c1 + d.c2), nb, label=label):
        if coeff is not None:
            buf, nbytes = _gn_buffer(B, hout * wout, cout, x.device) if want else (None, 0)
            slab = C.c_int32(0)
            check(lib.e2eft_conv2d_fwd_normed(C.byref(d), _ptr(x), C.c_void_p(coeff), _ptr(beta), 1 if silu else 0, _ptr(w_packed), _ptr(bias), _ptr(rowadd),
                                              _ptr(residual), _ptr(out), _ptr(buf), nbytes, C.byref(slab), _stream()))
            if want:
                _attach_stats(out, buf, slab.value, B, hout * wout, cout)
            out._e2eft_keep = ws            # (the coefficient workspace lives as long as the launch may: stream-ordered free after the output)
        elif sk:   # few output tiles, long reduction: split-K (the consumer GroupNorm computes its own statistics)
            ws = torch.empty(sk // x.element_size(), dtype=x.dtype, device=x.device)
            check(lib.e2eft_conv2d_fwd_splitk(C.byref(d), _ptr(x), _ptr(x2), _ptr(w_packed), _ptr(bias), _ptr(rowadd), _ptr(residual), _ptr(out),
                                              _ptr(ws), sk, _stream()))
        elif want:
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
    if (F32_SPLIT_ENABLED and a.dtype == torch.float32 and not want and not bias_along_m and (alpha == 1.0 or bias is None) and K % 64 == 0 and M > 256 and N % 8 == 0 and w.is_contiguous()
            and a.data_ptr() % 16 == 0 and d.lda % 4 == 0 and out.data_ptr() % 16 == 0 and (residual is None or residual.data_ptr() % 16 == 0)
            and (bias is None or bias.data_ptr() % 16 == 0)):
        # fp32 nn.Linear of whole 256-row tiles: two-term f16 split planes on the f16 matrix pipe (csrc/f32split.hip; igemm5's GEMM mode)
        ds = GemmDesc()
        ds.dtype = _lib.F32
        ds.m, ds.n, ds.k = M, N, K
        ds.lda, ds.ldw, ds.ldo, ds.ldr = 2 * K, 3 * K, d.ldo, d.ldr
        ds.nzo = ds.nzi = 1
        ds.bias_along_m = 0
        ds.alpha = alpha
        lib = _lib.load()
        if lib.e2eft_gemm_f32split_supported(C.byref(ds)) == 1:
            wsp, inv_sw = f32_split_weight(w, 1, K)
            planes = torch.empty((M, 2 * K), dtype=torch.float16, device=a.device)
            scale = torch.empty(4, dtype=torch.float32, device=a.device)
            with _timed("f32split", 0.0, 12.0 * M * K, label="split2 rows M%d K%d" % (M, K), launches=2):
                check(lib.e2eft_f32_split2(_ptr(a), M, K, d.lda, _ptr(planes), 2 * K, _ptr(scale), _stream()))
            with _timed("igemm", 6.0 * M * N * K, (M * 2 * K * 2 + N * 3 * K * 2 + M * N * 4 * (2 if residual is not None else 1)), label="gemm M%d N%d K%d f32split" % (M, N, K),
                        flops_nominal=2.0 * M * N * K):
                check(lib.e2eft_gemm_f32split(C.byref(ds), _ptr(planes), _ptr(scale), _ptr(wsp), _ptr(inv_sw), _ptr(bias), _ptr(residual), _ptr(out), _stream()))
            return out
    with _timed("igemm", 2.0 * M * N * K, (M * K + N * K + M * N * (2 if residual is not None else 1)) * a.element_size(), label="gemm M%d N%d K%d" % (M, N, K)):
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
    with _timed("igemm", 2.0 * m * n * k * nzo * nzi, float(nzo * nzi) * (m * k + n * k + m * n) * dtype_size(dtype), label="bgemm z%d M%d N%d K%d" % (nzo * nzi, m, n, k)):
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


NORM_FUSION_ENABLED = True    # conv2d(norm=...): GroupNorm(+SiLU) applied inside the consuming convolution where the library supports it


def groupnorm_stats(x, gamma, groups, eps):
    """The statistics half of groupnorm() alone -> (workspace, device address of the fp32 (a, mean) pairs [B][C][2] inside it)."""
    _check_cuda(x, gamma)
    B, H, W, c1 = x.shape
    d = GroupNormDesc()
    d.dtype = dtype_id(x.dtype)
    d.batch, d.hw = B, H * W
    d.c1, d.ldx1, d.c2, d.ldx2 = c1, _nhwc_ld(x), 0, 0
    d.groups, d.ldy, d.silu, d.eps = groups, c1, 0, eps
    lib = _lib.load()
    nbytes = lib.e2eft_groupnorm_workspace_bytes(C.byref(d))
    if nbytes == 0:
        raise RuntimeError("groupnorm_stats: %s" % lib.e2eft_last_error().decode())
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    s1 = getattr(x, "_e2eft_gn", None) if GN_STATS_ENABLED else None
    with _timed("groupnorm_stats", 0.0, 0.0, label="gn stats B%d %dx%d C%d" % (B, H, W, c1)):
        check(lib.e2eft_groupnorm_fwd_stats(C.byref(d), _ptr(x), C.c_void_p(0), _ptr(gamma), _ptr(s1.partial) if s1 else C.c_void_p(0), s1.nslabs if s1 else 0,
                                            C.c_void_p(0), 0, _ptr(ws), nbytes, _stream()))
    return ws, ws.data_ptr() + lib.e2eft_groupnorm_coeff_offset(C.byref(d))


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


def softmax_rows_(s, n, scale, causal_nq=0):
    """in-place softmax(scale * s[r, :n]) over a 2-D [rows, lds] contiguous buffer; pad columns [n, roundup) zeroed.
    causal_nq > 0: row r is query r % causal_nq and sees keys 0 .. r % causal_nq only."""
    _check_cuda(s)
    assert s.dim() == 2 and s.is_contiguous()
    if causal_nq:
        check(_lib.load().e2eft_softmax_rows_causal(dtype_id(s.dtype), s.shape[0], n, s.shape[1], scale, causal_nq, _ptr(s), _stream()))
    else:
        check(_lib.load().e2eft_softmax_rows(dtype_id(s.dtype), s.shape[0], n, s.shape[1], scale, _ptr(s), _stream()))
    return s


def _ld3(t):
    if t.shape[2] > 1 and t.stride(2) != 1:
        raise ValueError("attention operands need unit inner stride")
    ld = t.stride(1) if t.shape[1] > 1 else max(t.shape[2], t.stride(1))
    if t.shape[0] > 1 and t.stride(0) != t.shape[1] * ld:
        raise ValueError("attention operands must be batch-dense: %s %s" % (tuple(t.shape), t.stride()))
    return ld


def _attn_desc(q, k, v, out, heads, scale, kv_nseg, kv_bmod):
    B, Nq, Wd = q.shape
    Bkv, Nk, _ = k.shape
    assert Wd == heads * 64 and k.shape[2] == Wd and v.shape == k.shape
    d = AttnDesc()
    d.dtype = dtype_id(q.dtype)
    d.batch, d.heads, d.nq, d.nk_seg = B, heads, Nq, Nk
    d.kv_nseg = kv_nseg
    d.kv_bmod = B if kv_bmod is None else kv_bmod
    assert Bkv >= d.kv_bmod * kv_nseg
    d.ldq, d.ldk, d.ldv, d.ldo = _ld3(q), _ld3(k), _ld3(v), _ld3(out)
    d.scale = scale
    return d


def attention(q, k, v, heads, scale, kv_nseg=1, kv_bmod=None, out=None, return_lse=False):
    """Fused attention, head dim 64. q: [B,Nq,heads*64] view, k/v: [Bkv,Nk,heads*64] views (row-strided).
    return_lse: also return the [B, heads, Nq] fp32 log-sum-exp the fused backward needs."""
    _check_cuda(q, k, v, out)
    B, Nq, Wd = q.shape
    Nk = k.shape[1]
    if out is None:
        out = torch.empty((B, Nq, Wd), dtype=q.dtype, device=q.device)
    d = _attn_desc(q, k, v, out, heads, scale, kv_nseg, kv_bmod)
    lse = torch.empty((B, heads, Nq), dtype=torch.float32, device=q.device) if return_lse else None
    with _timed("attn", 4.0 * B * heads * Nq * Nk * kv_nseg * 64, label="attn B%d h%d Nq%d Nk%d" % (B, heads, Nq, Nk * kv_nseg)):
        check(_lib.load().e2eft_attn_fwd_lse(C.byref(d), _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), _stream()))
    return (out, lse) if return_lse else out


ATTN512_SPLIT_TAIL = True   # tests / A-B: without a workspace every query block runs all keys in one workgroup


def attention512(q, k, v, scale, out=None):
    """Fused attention of ONE 512-wide head (the VAE mid-block attention), fp16 / bf16.  q / k / v: [B, N, 512] views (row-strided: slices
    of one fused q|k|v projection are fine)."""
    _check_cuda(q, k, v, out)
    B, Nq, Wd = q.shape
    Nk = k.shape[1]
    assert Wd == 512 and k.shape[2] == 512 and v.shape == k.shape and k.shape[0] == B
    if out is None:
        out = torch.empty((B, Nq, 512), dtype=q.dtype, device=q.device)
    d = AttnDesc()
    d.dtype = dtype_id(q.dtype)
    d.batch, d.heads, d.nq, d.nk_seg, d.kv_nseg, d.kv_bmod = B, 1, Nq, Nk, 1, B
    d.ldq, d.ldk, d.ldv, d.ldo = _ld3(q), _ld3(k), _ld3(v), _ld3(out)
    d.scale = scale
    lib = _lib.load()
    nbytes = lib.e2eft_attn512_workspace_bytes(C.byref(d)) if ATTN512_SPLIT_TAIL else 0
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=q.device) if nbytes else None
    with _timed("attn512", 4.0 * B * Nq * Nk * 512, label="attn512 B%d Nq%d Nk%d" % (B, Nq, Nk)):
        check(lib.e2eft_attn512_fwd(C.byref(d), _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(ws), nbytes, _stream()))
    return out


def attention_bwd(q, k, v, out, dout, lse, heads, scale, dq, dk, dv):
    """Fused attention backward (head dim 64; fp16 / bf16 / strict fp32): writes dq / dk / dv (row-strided views shaped like q / k / v)."""
    _check_cuda(q, k, v, out, dout, lse, dq, dk, dv)
    B, Nq, Wd = q.shape
    Nk = k.shape[1]
    d = _attn_desc(q, k, v, out, heads, scale, 1, None)
    assert dout.shape == out.shape and dq.shape == q.shape and dk.shape == k.shape and dv.shape == v.shape
    assert lse.shape == (B, heads, Nq) and lse.dtype == torch.float32 and lse.is_contiguous()
    lib = _lib.load()
    nbytes = lib.e2eft_attn_bwd_workspace_bytes(C.byref(d))
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=q.device)
    with _timed("attn_bwd", 14.0 * B * heads * Nq * Nk * 64, label="attn_bwd B%d h%d Nq%d Nk%d" % (B, heads, Nq, Nk)):
        check(lib.e2eft_attn_bwd(C.byref(d), _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(dout), _ld3(dout), _ptr(lse), _ptr(dq), _ld3(dq), _ptr(dk), _ld3(dk),
                                 _ptr(dv), _ld3(dv), _ptr(ws), nbytes, _stream()))


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


ACT_KINDS = {"quick_gelu": 0, "gelu": 1, "silu": 2, "sigmoid": 3}


def activation(x, kind):
    """y = act(x) elementwise; kind in ACT_KINDS"""
    _check_cuda(x)
    x = x.contiguous()
    out = torch.empty_like(x)
    check(_lib.load().e2eft_activation(dtype_id(x.dtype), ACT_KINDS[kind], x.numel(), _ptr(x), _ptr(out), _stream()))
    return out


def depth_head(x, to_unit, dtype=None):
    """decoder output NHWC [B,H,W,>=3 (view)] -> [B,1,H,W]: to_unit False clip(mean_c, -1, 1); True the same mapped to [0,1];
    "mean" the bare channel mean (no clip)"""
    _check_cuda(x)
    B, H, W, _ = x.shape
    dtype = dtype or x.dtype
    y = torch.empty((B, 1, H, W), dtype=dtype, device=x.device)
    mode = 2 if to_unit == "mean" else (1 if to_unit else 0)
    check(_lib.load().e2eft_depth_head(dtype_id(x.dtype), dtype_id(dtype), B * H * W, _nhwc_ld(x), mode, _ptr(x), _ptr(y), _stream()))
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


# =========================================================================================================
# backward-pass kernels (include/e2eft.h "Backward pass of the E2E-FT training step"); used by autograd.py
def _conv_desc(x, x2, cout, kh, kw, stride, pad, up_to, alpha, ldo=0):
    B, H, W, c1 = x.shape
    hl, wl = (H, W) if up_to is None else up_to
    pt, pb, pl, pr = pad
    d = ConvDesc()
    d.dtype = dtype_id(x.dtype)
    d.batch, d.hin, d.win, d.hl, d.wl = B, H, W, hl, wl
    d.c1, d.ldx1 = c1, _nhwc_ld(x)
    d.c2, d.ldx2 = (x2.shape[3], _nhwc_ld(x2)) if x2 is not None else (0, 0)
    d.kh, d.kw, d.stride, d.pad_t, d.pad_l = kh, kw, stride, pt, pl
    d.hout = (hl + pt + pb - kh) // stride + 1
    d.wout = (wl + pl + pr - kw) // stride + 1
    d.cout, d.ldo, d.ldr, d.ldw = cout, ldo, 0, 0
    d.alpha = alpha
    return d


def transpose(x, rows_pad=None, out=None):
    """x: [Z, R, C] (or [R, C]) row-strided view -> [Z, C, rows_pad] with zeros in [R, rows_pad) (rows_pad multiple of 64 by default
    so that the result is a FAST-path K operand of e2eft_gemm)."""
    _check_cuda(x, out)
    squeeze = x.dim() == 2
    if squeeze:
        x = x[None]
    Z, R, Cc = x.shape
    if Cc > 1 and x.stride(2) != 1:
        raise ValueError("transpose: unit inner stride required")
    ld = x.stride(1) if R > 1 else max(Cc, x.stride(1))
    bs = x.stride(0) if Z > 1 else R * ld
    rp = round_up(R, 64) if rows_pad is None else rows_pad
    if out is None:
        out = torch.empty((Z, Cc, rp), dtype=x.dtype, device=x.device)
    assert out.shape == (Z, Cc, rp) and out.is_contiguous()
    with _timed("transpose", 0.0, 2.0 * Z * R * Cc * x.element_size(), label="transpose z%d %dx%d" % (Z, R, Cc)):
        check(_lib.load().e2eft_transpose(dtype_id(x.dtype), Z, R, Cc, ld, bs, rp, rp, Cc * rp, _ptr(x), _ptr(out), _stream()))
    return out[0] if squeeze else out


def dgrad_weight_from_packed(pk, cout, taps, cp):
    """The dgrad operand of a convolution from its packed forward weight: pk [cout, taps*cp] (OHWI rows, 16-byte channel padding) ->
    [cp, taps*cop] with out[ci][(taps-1-t)*cop + co] = pk[co][t*cp + ci], cop = cout padded to a 16-byte multiple with zeros (include/e2eft.h,
    e2eft_conv2d_dgrad: flipped taps, channels transposed).  ONE batched e2eft_transpose: batch = tap, the output walks the taps backwards
    (negative batch stride)."""
    _check_cuda(pk)
    e = epc(pk.dtype)
    assert pk.dim() == 2 and pk.shape == (cout, taps * cp) and pk.is_contiguous() and cp % e == 0 and pk.data_ptr() % 16 == 0
    cop = round_up(cout, e)
    out = torch.empty((cp, taps * cop), dtype=pk.dtype, device=pk.device)
    last = out.data_ptr() + (taps - 1) * cop * pk.element_size()
    with _timed("transpose", 0.0, 2.0 * taps * cout * cp * pk.element_size(), label="dgrad weight %dx%d taps%d" % (cout, cp, taps)):
        check(_lib.load().e2eft_transpose(dtype_id(pk.dtype), taps, cout, cp, taps * cp, cp, cop, taps * cop, -cop, _ptr(pk), C.c_void_p(last), _stream()))
    return out


def splitk_plan(M, N, K, dtype=None, cus=256):
    """(nsplit, kc): weight-gradient GEMMs are [Cout x kh*kw*Cin] outputs contracted over 10^4..10^5 pixels — a handful of output
    tiles.  Split the contraction into nsplit chunks of kc (multiple of 64) columns so that >= ~512 workgroups exist; the
    operands are zero padded to nsplit*kc columns by their producers (transpose / im2col_t).
    fp32 (round 5): the batched GEMM runs on igemm2's 8-wave kernel — 256 x 128 tiles, ONE workgroup per CU — so what matters is whole ROUNDS of `cus`
    tiles: nsplit is the chunk count (chunks of >= 512 columns) whose tile total fills its last round best (z9 x 30 tiles = 270 on 256 CUs ran two rounds at
    53 %; 8 or 17 chunks fill theirs to 94 / 99.6 %), the smallest such count among near-ties (fewer partials for the reduction pass)."""
    if dtype == torch.float32:
        tiles = ((M + 255) // 256) * ((N + 127) // 128)
        if tiles * 1 >= cus // 2 and K < 1024:
            return 1, round_up(K, 64)
        best = None
        for ns in range(1, max(1, K // 512) + 1):
            tot = tiles * ns
            if tot > 8 * cus and best is not None:
                break
            rounds = (tot + cus - 1) // cus
            eff = tot / (rounds * cus)
            if best is None or eff > best[0] + 0.02:
                best = (eff, ns)
        want = best[1]
        kc = round_up((K + want - 1) // want, 64)
        return (K + kc - 1) // kc, kc
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    want = max(1, min((512 + tiles - 1) // tiles, K // 512))
    kc = round_up((K + want - 1) // want, 64)
    nsplit = (K + kc - 1) // kc
    return nsplit, kc


def gemm_splitk(a, w, nsplit, kc, alpha=1.0):
    """out[m,n] = alpha * sum_k a[m,k] w[n,k] for a [M, nsplit*kc], w [N, nsplit*kc]: nsplit batched partial GEMMs + a column-sum
    reduction in fp32.  Returns fp32 [M, N] (or the plain GEMM result in the operand dtype when nsplit == 1)."""
    if nsplit == 1:
        return gemm(a, w, alpha=alpha)
    M, K = a.shape
    N = w.shape[0]
    assert K == nsplit * kc and w.shape[1] == K and N % epc(a.dtype) == 0
    part = torch.empty((nsplit, M, N), dtype=a.dtype, device=a.device)
    bgemm_raw(a.dtype, M, N, kc, a, _rows_ld(a), (0, kc), w, _rows_ld(w), (0, kc), part, N, (0, M * N), 1, nsplit)
    return colsum(part.view(nsplit, M * N), groups=1, alpha=alpha).view(M, N)


def im2col_t(x, x2, kh, kw, stride, pad, up_to=None, Pp=None):
    """-> ([kh*kw*(c1+c2), Ppad] K-contiguous im2col of the conv input over the OUTPUT pixels, P, Ppad)"""
    _check_cuda(x, x2)
    d = _conv_desc(x, x2, 1, kh, kw, stride, pad, up_to, 1.0)
    P = d.batch * d.hout * d.wout
    Pp = round_up(P, 64) if Pp is None else Pp
    cin = d.c1 + d.c2
    col = torch.empty((kh * kw * cin, Pp), dtype=x.dtype, device=x.device)
    with _timed("im2col_t", 0.0, 2.0 * kh * kw * cin * Pp * x.element_size(), label="im2col_t %dx%d %d P%d" % (kh, kw, cin, P)):
        check(_lib.load().e2eft_conv2d_im2col_t(C.byref(d), _ptr(x), _ptr(x2), _ptr(col), Pp, _stream()))
    return col, P, Pp


WGRAD_DIRECT = True   # tests / A-B: False keeps every weight gradient on the transpose + im2col_t + split-K GEMM path
WGRAD_DIRECT_FP32 = True   # round 6: strict-fp32 weight gradients straight from the NHWC tensors too (csrc/wgrad.hip::wgrad32_kernel); False: the round-5 path


def conv2d_wgrad(dy, x, x2, cout, kh, kw, stride, pad, alpha, out=None):
    """Weight gradient straight from the NHWC tensors (csrc/wgrad.hip): dy [B,hout,wout,>=cout] (pixel-dense), x [B,H,W,c1], x2 optional second
    concat source -> fp32 [cout, kh*kw*(c1+c2)] (OHWI rows), or None when the kernel does not serve the problem (the caller falls back).
    out: a dense fp32 tensor of cout * kh*kw*(c1+c2) elements that receives the result (FlatAdamW's gradient slot: the reduction's last pass — or the kernel
    itself when it does not split — writes there, no copy / add afterwards).
    The kernel addresses each operand through ONE 32-bit buffer descriptor, so tensors of 4 GB and more (configs[2] as benchmarked: 32 images of
    576^2 x 256 channels = 5.4 GB) are cut along the batch into launches below that limit; every launch adds its split partials to the same reduction."""
    if not WGRAD_DIRECT or (dy.dtype == torch.float32 and not WGRAD_DIRECT_FP32):
        return None
    _check_cuda(dy, x, x2)
    cin_all = x.shape[3] + (x2.shape[3] if x2 is not None else 0)
    if (dy.dtype == torch.float32 and F32_SPLIT_ENABLED and WGRAD_F32_SPLIT and cin_all % 64 == 0 and x.shape[3] % 8 == 0 and cout % 64 == 0 and dy.shape[3] % 8 == 0
            and _lib.load().e2eft_get_option(_lib.OPT_F32_SPLIT) == 1 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
            and (x2 is None or (x2.shape[3] % 8 == 0 and x2.data_ptr() % 16 == 0))):
        # fp32 on the f16 matrix pipe (csrc/f32split.hip): dy s_dy = d0 + d1, x s_x = x0 + x1 (two-term f16 splits, exact to 2^-22); the gradient is the sum of the
        # 16-bit kernel's results for (d0, x0), (d0, x1), (d1, x0), scaled back by the two device scalars — three launches at the f16 rate instead of one at the fp32 rate
        dyp, sdy = f32_split2(dy, keep=True)          # (the data-gradient convolution of the same dY has usually split it already: autograd._Conv2dFn.backward)
        xp, sx = f32_split2(x, x2=x2)                 # (two sources: the planes of their concatenation — the 16-bit kernel then sees ONE source of c1 + c2 channels)
        c0, c1 = dy.shape[3], cin_all
        r = None
        for (a_, b_) in ((dyp[..., :c0], xp[..., :c1]), (dyp[..., :c0], xp[..., c1:]), (dyp[..., c0:], xp[..., :c1])):
            t = conv2d_wgrad(a_, b_, None, cout, kh, kw, stride, pad, alpha)
            if t is None:
                r = None
                break
            r = t if r is None else r.add_(t)
        if r is not None:
            r = r.mul_(sdy[2] * sx[2])
            if out is not None:
                out.view(r.shape).copy_(r)
                return out.view(r.shape)
            return r
    B = x.shape[0]
    lddy = _nhwc_ld(dy)
    es = dy.element_size()
    per_img = max(dy.shape[1] * dy.shape[2] * lddy, x.shape[1] * x.shape[2] * max(_nhwc_ld(x), _nhwc_ld(x2) if x2 is not None else 0)) * es
    pix_img = max(dy.shape[1] * dy.shape[2], 1)
    step = max(1, min(B, (0xFFFF0000 - 1) // max(per_img, 1), ((1 << 24) - 1) // pix_img))
    lib = _lib.load()
    parts = []
    cin = x.shape[3] + (x2.shape[3] if x2 is not None else 0)
    N = kh * kw * cin
    for b0 in range(0, B, step):
        b1 = min(B, b0 + step)
        xs, x2s, dys = x[b0:b1], (x2[b0:b1] if x2 is not None else None), dy[b0:b1]
        d = _conv_desc(xs, x2s, cout, kh, kw, stride, pad, None, alpha)
        assert tuple(dys.shape[:3]) == (d.batch, d.hout, d.wout), (dys.shape, d.batch, d.hout, d.wout)
        nbytes = lib.e2eft_conv2d_wgrad_workspace_bytes(C.byref(d), lddy)
        if nbytes == 0:
            return None
        direct = out is not None and step >= B and nbytes == cout * N * 4      # one launch, no split: the kernel's workspace IS the result
        part = out.view(-1) if direct else torch.empty(nbytes // 4, dtype=torch.float32, device=dy.device)
        ns = C.c_int32(0)
        P = d.batch * d.hout * d.wout
        with _timed("wgrad", 2.0 * P * cout * N, (P * (cout + cin)) * es, label="wgrad %dx%ds%d P%d %d->%d" % (kh, kw, stride, P, cin, cout)):
            check(lib.e2eft_conv2d_wgrad(C.byref(d), _ptr(dys), lddy, _ptr(xs), _ptr(x2s), _ptr(part), nbytes, C.byref(ns), _stream()))
        parts.append(part.view(ns.value, cout * N))
    if len(parts) == 1 and parts[0].shape[0] == 1:
        if out is not None and parts[0].data_ptr() != out.data_ptr():
            out.view(1, -1).copy_(parts[0])
            return out.view(cout, N)
        return parts[0].view(cout, N)
    allp = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
    return colsum(allp, groups=1, out=None if out is None else out.view(1, -1)).view(cout, N)


def linear_wgrad(dy2d, x2d, alpha=1.0, out=None):
    """dW [N, K] fp32 = alpha * dy2d^T x2d for row views dy2d [M, N], x2d [M, K] (K a multiple of 64) — the 1x1 case of conv2d_wgrad; None = fall back.
    out: see conv2d_wgrad"""
    if not WGRAD_DIRECT or (dy2d.dtype == torch.float32 and not WGRAD_DIRECT_FP32) or x2d.shape[1] % 64 != 0:
        return None
    M, N = dy2d.shape
    K = x2d.shape[1]
    try:
        ldy, ldx = _rows_ld(dy2d), _rows_ld(x2d)
    except ValueError:
        return None
    if ldy % epc(dy2d.dtype) != 0 or ldx % epc(dy2d.dtype) != 0 or dy2d.data_ptr() % 16 != 0 or x2d.data_ptr() % 16 != 0:
        return None
    xv = x2d.as_strided((1, 1, M, K), (M * ldx, M * ldx, ldx, 1))
    dv = dy2d.as_strided((1, 1, M, N), (M * ldy, M * ldy, ldy, 1))
    return conv2d_wgrad(dv, xv, None, N, 1, 1, 1, (0, 0, 0, 0), alpha, out=out)


def conv2d_dgrad(dy, w_dgrad, x_shape, c2, kh, kw, stride, pad, up_to, alpha):
    """dy [B,hout,wout,cout_pad] (zero / finite pad channels), w_dgrad [cin, kh*kw*cout_pad]; x_shape = forward x [B,H,W,c1].
    Returns the gradient w.r.t. the logical (upsampled) concatenated input [B, hl, wl, c1+c2]."""
    _check_cuda(dy, w_dgrad)
    B, H, W, c1 = x_shape
    hl, wl = (H, W) if up_to is None else up_to
    pt, pb, pl, pr = pad
    d = ConvDesc()
    d.dtype = dtype_id(dy.dtype)
    d.batch, d.hin, d.win, d.hl, d.wl = B, H, W, hl, wl
    d.c1, d.ldx1, d.c2, d.ldx2 = c1, c1, c2, c2
    d.kh, d.kw, d.stride, d.pad_t, d.pad_l = kh, kw, stride, pt, pl
    d.hout, d.wout = dy.shape[1], dy.shape[2]
    assert d.hout == (hl + pt + pb - kh) // stride + 1 and d.wout == (wl + pl + pr - kw) // stride + 1
    cop = dy.shape[3]
    d.cout, d.alpha = cop, alpha
    cin = c1 + c2
    assert w_dgrad.shape == (cin, kh * kw * cop) and w_dgrad.is_contiguous() and w_dgrad.dtype == dy.dtype
    label = "dgrad%dx%ds%d B%d %dx%d %d->%d" % (kh, kw, stride, B, hl, wl, cop, cin)
    if (F32_SPLIT_ENABLED and dy.dtype == torch.float32 and ((kh, kw, stride, tuple(pad)) == (3, 3, 1, (1, 1, 1, 1)) or (kh, kw, stride, tuple(pad)) == (1, 1, 1, (0, 0, 0, 0)))
            and up_to is None and cop % 64 == 0 and cin % 8 == 0):
        # the data gradient of a 3x3 / stride-1 / pad-1 (or 1x1) convolution IS such a convolution of dY with the flipped, transposed weights: the f16-split route of conv2d
        dx = new_nhwc(B, hl, wl, cin, dy.dtype, dy.device)
        if _conv2d_f32split(dy, w_dgrad, None, cin, None, alpha, dx, False, label, keep_planes=True, geom=(kh, kw, 1, tuple(pad))) is not None:
            return dx
    else:
        dx = new_nhwc(B, hl, wl, cin, dy.dtype, dy.device)
    with _timed("igemm", 2.0 * B * hl * wl * cin * kh * kw * cop, label=label):
        check(_lib.load().e2eft_conv2d_dgrad(C.byref(d), _ptr(dy), _nhwc_ld(dy), cop, _ptr(w_dgrad), w_dgrad.shape[1], _ptr(dx), _nhwc_ld(dx),
                                             _stream()))
    return dx


def colsum(x2d, groups=1, alpha=1.0, out=None):
    """x2d [rows, cols] row-strided -> fp32 [groups, cols] sums over each group of rows/groups consecutive rows (into `out`, dense fp32 [groups, cols], if given)"""
    _check_cuda(x2d, out)
    R, Cc = x2d.shape
    assert R % groups == 0
    lib = _lib.load()
    nbytes = lib.e2eft_colsum_workspace_bytes(groups, R // groups, Cc)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x2d.device)
    if out is None:
        out = torch.empty((groups, Cc), dtype=torch.float32, device=x2d.device)
    else:
        assert out.dtype == torch.float32 and tuple(out.shape) == (groups, Cc) and out.is_contiguous(), (out.dtype, out.shape, out.stride())
    check(lib.e2eft_colsum(dtype_id(x2d.dtype), groups, R // groups, Cc, _rows_ld(x2d), alpha, _ptr(x2d), _ptr(out), _ptr(ws), nbytes, _stream()))
    return out


def upsample_nearest_bwd(dy, H, W):
    """dy [B,hl,wl,C] -> [B,H,W,C]"""
    _check_cuda(dy)
    B, hl, wl, Cc = dy.shape
    dx = torch.empty((B, H, W, Cc), dtype=dy.dtype, device=dy.device)
    check(_lib.load().e2eft_upsample_nearest_bwd(dtype_id(dy.dtype), B, H, W, hl, wl, Cc, _nhwc_ld(dy), Cc, _ptr(dy), _ptr(dx), _stream()))
    return dx


def groupnorm_fwd_ws(x, gamma, beta, groups, eps, silu=False, x2=None, s1=None, s2=None):
    """GroupNorm forward that also returns its workspace (mean / rstd live there) for groupnorm_bwd.  s1 / s2: GnStats of x / x2
    emitted by their producers (statistics pass skipped for that source)."""
    _check_cuda(x, gamma, beta, x2)
    B, H, W, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[3]
    out = torch.empty((B, H, W, c1 + c2), dtype=x.dtype, device=x.device)
    d = _gn_desc(x, x2, groups, eps, silu, _nhwc_ld(out))
    lib = _lib.load()
    nbytes = lib.e2eft_groupnorm_workspace_bytes(C.byref(d))
    if nbytes == 0:
        raise RuntimeError("groupnorm: %s" % lib.e2eft_last_error().decode())
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    if not GN_STATS_ENABLED:
        s1 = s2 = None
    with _timed("groupnorm", 0.0, 2.0 * B * H * W * (c1 + c2) * x.element_size(), label="gn B%d %dx%d C%d" % (B, H, W, c1 + c2)):
        check(lib.e2eft_groupnorm_fwd_pre(C.byref(d), _ptr(x), _ptr(x2), _ptr(gamma), _ptr(beta), _ptr(out),
                                          _ptr(s1.partial) if s1 else C.c_void_p(0), s1.nslabs if s1 else 0,
                                          _ptr(s2.partial) if (s2 and x2 is not None) else C.c_void_p(0), s2.nslabs if (s2 and x2 is not None) else 0,
                                          _ptr(ws), nbytes, _stream()))
    return out, ws


def _gn_desc(x, x2, groups, eps, silu, ldy):
    B, H, W, c1 = x.shape
    d = GroupNormDesc()
    d.dtype = dtype_id(x.dtype)
    d.batch, d.hw = B, H * W
    d.c1, d.ldx1 = c1, _nhwc_ld(x)
    d.c2, d.ldx2 = (x2.shape[3], _nhwc_ld(x2)) if x2 is not None else (0, 0)
    d.groups, d.ldy, d.silu, d.eps = groups, ldy, 1 if silu else 0, eps
    return d


def groupnorm_bwd(x, x2, gamma, beta, groups, eps, silu, dy, fwd_ws, need_dx=True, need_dparams=True, dx_add=None, dg_out=None, db_out=None):
    """-> (dx [B,H,W,C] or None, dgamma fp32 [C] or None, dbeta fp32 [C] or None); dx_add: a [B,H,W,C] gradient added into dx;
    dg_out / db_out: dense fp32 [C] tensors that receive dgamma / dbeta"""
    _check_cuda(x, x2, gamma, beta, dy, fwd_ws)
    B, H, W, c1 = x.shape
    Cc = c1 + (0 if x2 is None else x2.shape[3])
    d = _gn_desc(x, x2, groups, eps, silu, Cc)
    lib = _lib.load()
    nbytes = lib.e2eft_groupnorm_bwd_workspace_bytes(C.byref(d))
    if nbytes == 0:
        raise RuntimeError("groupnorm_bwd: %s" % lib.e2eft_last_error().decode())
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    dx = torch.empty((B, H, W, Cc), dtype=x.dtype, device=x.device) if need_dx else None
    for o_ in (dg_out, db_out):
        assert o_ is None or (o_.dtype == torch.float32 and tuple(o_.shape) == (Cc,) and o_.is_contiguous())
    dg = (dg_out if dg_out is not None else torch.empty(Cc, dtype=torch.float32, device=x.device)) if need_dparams else None
    db = (db_out if db_out is not None else torch.empty(Cc, dtype=torch.float32, device=x.device)) if need_dparams else None
    with _timed("groupnorm_bwd", 0.0, 5.0 * B * H * W * Cc * x.element_size(), label="gn_bwd B%d %dx%d C%d" % (B, H, W, Cc)):
        check(lib.e2eft_groupnorm_bwd_add(C.byref(d), _ptr(x), _ptr(x2), _ptr(gamma), _ptr(beta), _ptr(dy), _nhwc_ld(dy), _ptr(dx_add),
                                          _nhwc_ld(dx_add) if dx_add is not None else 0, _ptr(dx), Cc, _ptr(dg), _ptr(db), _ptr(fwd_ws), _ptr(ws), nbytes,
                                          _stream()))
    return dx, dg, db


def layernorm_bwd(x, gamma, eps, dy, need_dx=True, gb_out=None):
    """-> (dx like x or None, dgamma fp32 [C], dbeta fp32 [C]); gb_out: dense fp32 [2, C] that receives (dgamma, dbeta)"""
    _check_cuda(x, gamma, dy)
    Cc = x.shape[-1]
    a = x.reshape(-1, Cc) if x.is_contiguous() else _as_rows(x)
    g = dy.reshape(-1, Cc) if dy.is_contiguous() else _as_rows(dy)
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device) if need_dx else None
    lib = _lib.load()
    nbytes = lib.e2eft_layernorm_bwd_workspace_bytes(a.shape[0], Cc)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    if gb_out is not None:
        assert gb_out.dtype == torch.float32 and tuple(gb_out.shape) == (2, Cc) and gb_out.is_contiguous()
    gb = gb_out if gb_out is not None else torch.empty((2, Cc), dtype=torch.float32, device=x.device)
    check(lib.e2eft_layernorm_bwd(dtype_id(x.dtype), a.shape[0], Cc, _rows_ld(a), _rows_ld(g), Cc, eps, _ptr(a), _ptr(gamma), _ptr(g), _ptr(dx), _ptr(gb),
                                  _ptr(ws), nbytes, _stream()))
    return dx, gb[0], gb[1]


def geglu_bwd(h, dy):
    _check_cuda(h, dy)
    c2 = h.shape[-1]
    c = c2 // 2
    a = h.reshape(-1, c2) if h.is_contiguous() else _as_rows(h)
    g = dy.reshape(-1, c) if dy.is_contiguous() else _as_rows(dy)
    dh = torch.empty(h.shape, dtype=h.dtype, device=h.device)
    check(_lib.load().e2eft_geglu_bwd(dtype_id(h.dtype), a.shape[0], c, _rows_ld(a), _rows_ld(g), c2, _ptr(a), _ptr(g), _ptr(dh), _stream()))
    return dh


def softmax_bwd_rows_(p, dp, n, scale):
    """in place on dp [rows, lds]: dp <- p * (dp - rowsum(dp*p)) * scale"""
    _check_cuda(p, dp)
    assert p.dim() == 2 and p.is_contiguous() and dp.shape == p.shape and dp.is_contiguous() and p.dtype == dp.dtype
    check(_lib.load().e2eft_softmax_bwd_rows(dtype_id(p.dtype), p.shape[0], n, p.shape[1], scale, _ptr(p), _ptr(dp), _stream()))
    return dp


def silu_bwd(x, dy):
    _check_cuda(x, dy)
    x, dy = x.contiguous(), dy.contiguous()
    dx = torch.empty_like(x)
    check(_lib.load().e2eft_silu_bwd(dtype_id(x.dtype), x.numel(), _ptr(x), _ptr(dy), _ptr(dx), _stream()))
    return dx


def depth_head_bwd(x, dy, to_unit):
    """x NHWC [B,H,W,>=3] (forward input), dy [B,1,H,W] -> dx [B,H,W,3] (view of a zero-padded buffer)"""
    _check_cuda(x, dy)
    B, H, W, _ = x.shape
    dy = dy.contiguous()
    cp = round_up(3, epc(x.dtype))
    dx = torch.empty((B, H, W, cp), dtype=x.dtype, device=x.device)
    check(_lib.load().e2eft_depth_head_bwd(dtype_id(x.dtype), dtype_id(dy.dtype), B * H * W, _nhwc_ld(x), cp, cp, 1 if to_unit else 0, _ptr(x), _ptr(dy),
                                           _ptr(dx), _stream()))
    return dx[..., :x.shape[3]] if x.shape[3] <= cp else dx


def normal_head_bwd(x, dy, clamp, sign):
    _check_cuda(x, dy)
    B, H, W, _ = x.shape
    dy = dy.contiguous()
    cp = round_up(3, epc(x.dtype))
    dx = torch.empty((B, H, W, cp), dtype=x.dtype, device=x.device)
    check(_lib.load().e2eft_normal_head_bwd(dtype_id(x.dtype), dtype_id(dy.dtype), B, H * W, _nhwc_ld(x), cp, cp, 1 if clamp else 0, sign, _ptr(x), _ptr(dy),
                                            _ptr(dx), _stream()))
    return dx[..., :x.shape[3]] if x.shape[3] <= cp else dx


def ssi_loss_fwd_saved(p, t, m):
    """p, t fp32 [B, hw] contiguous, m uint8 [B, hw] -> (loss[1], scale_shift [B,2], workspace) for ssi_loss_bwd"""
    lib = _lib.load()
    B = p.shape[0]
    nbytes = lib.e2eft_ssi_loss_workspace_bytes(B)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=p.device)
    out = torch.empty(1, dtype=torch.float32, device=p.device)
    ss = torch.empty((B, 2), dtype=torch.float32, device=p.device)
    check(lib.e2eft_ssi_loss_fwd(B, p.shape[1], _ptr(p), _ptr(t), _ptr(m), _ptr(out), _ptr(ss), _ptr(ws), nbytes, _stream()))
    return out, ss, ws


def ssi_loss_bwd(p, t, m, ss, fwd_ws, gout):
    B = p.shape[0]
    dp = torch.empty_like(p)
    ws = torch.empty(2 * B, dtype=torch.float64, device=p.device)
    g = gout.reshape(1).float().contiguous()
    check(_lib.load().e2eft_ssi_loss_bwd(B, p.shape[1], _ptr(p), _ptr(t), _ptr(m), _ptr(ss), _ptr(fwd_ws), _ptr(g), _ptr(dp), _ptr(ws), 16 * B, _stream()))
    return dp


def angular_loss_fwd_saved(p, t, m):
    """p, t fp32 [B,3,hw], m uint8 [B,hw]"""
    lib = _lib.load()
    ws = torch.empty(2, dtype=torch.float64, device=p.device)
    out = torch.empty(1, dtype=torch.float32, device=p.device)
    check(lib.e2eft_angular_loss_fwd(p.shape[0], p.shape[2], _ptr(p), _ptr(t), _ptr(m), _ptr(out), _ptr(ws), 16, _stream()))
    return out, ws


def angular_loss_bwd(p, t, m, fwd_ws, gout):
    dp = torch.empty_like(p)
    g = gout.reshape(1).float().contiguous()
    check(_lib.load().e2eft_angular_loss_bwd(p.shape[0], p.shape[2], _ptr(p), _ptr(t), _ptr(m), _ptr(fwd_ws), _ptr(g), _ptr(dp), _stream()))
    return dp


def sumsq(g, out=None):
    """fp64 [1] device scalar = sum g^2 over a flat fp32 buffer"""
    _check_cuda(g)
    assert g.dtype == torch.float32 and g.is_contiguous()
    if out is None:
        out = torch.empty(1, dtype=torch.float64, device=g.device)
    check(_lib.load().e2eft_sumsq(g.numel(), _ptr(g), _ptr(out), _stream()))
    return out


def adamw_step_(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_sumsq=None, grad_scale=1.0, max_norm=0.0):
    _check_cuda(param, grad, exp_avg, exp_avg_sq, grad_sumsq)
    for t in (param, grad, exp_avg, exp_avg_sq):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == param.numel()
    check(_lib.load().e2eft_adamw_step(param.numel(), _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), lr, beta1, beta2, eps, weight_decay, step,
                                       _ptr(grad_sumsq), grad_scale, max_norm, _stream()))
    return param


def adamw_step_guarded_(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, state, coef, grad_sumsq, grad_scale=1.0, max_norm=0.0):
    """AdamW with the step counter on the device (`state` int64 [2] = applied / skipped steps) and a skip on a non-finite gradient norm"""
    _check_cuda(param, grad, exp_avg, exp_avg_sq, state, coef, grad_sumsq)
    for t in (param, grad, exp_avg, exp_avg_sq):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == param.numel()
    assert state.dtype == torch.int64 and state.numel() == 2 and coef.dtype == torch.float32 and coef.numel() == 4 and grad_sumsq.dtype == torch.float64
    check(_lib.load().e2eft_adamw_step_guarded(param.numel(), _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), lr, beta1, beta2, eps, weight_decay,
                                               _ptr(state), _ptr(coef), _ptr(grad_sumsq), grad_scale, max_norm, _stream()))
    return param


def ema_step_(shadow, param, one_minus_decay):
    """shadow -= one_minus_decay * (shadow - param) over flat fp32 buffers (diffusers EMAModel.step; csrc/bwd.hip ema_kernel)"""
    _check_cuda(shadow, param)
    assert shadow.dtype == param.dtype == torch.float32 and shadow.is_contiguous() and param.is_contiguous() and shadow.numel() == param.numel()
    check(_lib.load().e2eft_ema_step(shadow.numel(), _ptr(shadow), _ptr(param), float(one_minus_decay), _stream()))
    return shadow


def cast_(x, y, mul=1.0, accumulate=False):
    """y <- (y if accumulate else 0) + x*mul with dtype conversion; flat contiguous buffers"""
    _check_cuda(x, y)
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    check(_lib.load().e2eft_cast(dtype_id(x.dtype), dtype_id(y.dtype), x.numel(), mul, 1 if accumulate else 0, _ptr(x), _ptr(y), _stream()))
    return y


# ---------------------------------------------------------------------------------------------------------
# pre- / post-processing of the pipelines' __call__ (csrc/prepost.hip)
def resample_bilinear_aa(x, size, xtab, ytab, round_u8=False, mul=1.0, add=0.0):
    """torch's antialiased bilinear resize of planar images: x [P, h0, w0] uint8 or fp32 (device, contiguous) -> fp32 [P, h, w];
    xtab / ytab = (bounds int32 [out, 2], weights fp32 [out, ksize]) device tables (pipeline.aa_bilinear_tables); round_u8: round to the
    uint8 grid first; then v * mul + add."""
    _check_cuda(x, xtab[0], xtab[1], ytab[0], ytab[1])
    assert x.dim() == 3 and x.is_contiguous() and x.dtype in (torch.uint8, torch.float32)
    P, h0, w0 = x.shape
    h, w = size
    mid = torch.empty((P, h0, w), dtype=torch.float32, device=x.device)
    out = torch.empty((P, h, w), dtype=torch.float32, device=x.device)
    check(_lib.load().e2eft_resample_bilinear_aa(P, h0, w0, h, w, 1 if x.dtype == torch.uint8 else 0, _ptr(x), _ptr(xtab[0]), _ptr(xtab[1]), xtab[1].shape[1],
                                                 _ptr(ytab[0]), _ptr(ytab[1]), ytab[1].shape[1], 1 if round_u8 else 0, mul, add, _ptr(mid), _ptr(out), _stream()))
    return out


def minmax_unit(x):
    """(x - min) / (max - min) over the whole fp32 tensor (zeros when max == min)"""
    _check_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    lib = _lib.load()
    nbytes = lib.e2eft_minmax_unit_workspace_bytes()
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    out = torch.empty_like(x)
    check(lib.e2eft_minmax_unit(x.numel(), _ptr(x), _ptr(out), _ptr(None), _ptr(ws), nbytes, _stream()))
    return out


# ---------------------------------------------------------------------------------------------------------
# test-time ensembling (csrc/ensemble.hip): x is the fp32 [N, ...] stack of the N predictions of one image
def _ens_stack(x):
    _check_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() >= 2, "ensemble: need a contiguous fp32 [N, ...] stack"
    n = x.shape[0]
    lib = _lib.load()
    nbytes = lib.e2eft_ensemble_workspace_bytes(n)
    if nbytes == 0:
        raise RuntimeError("ensemble: %s" % lib.e2eft_last_error().decode())
    return n, x.numel() // n, torch.empty(nbytes // 8, dtype=torch.float64, device=x.device), nbytes


def ensemble_minmax(x):
    """[N, 2] (min, max) of every image of the stack"""
    n, npix, ws, nbytes = _ens_stack(x)
    out = torch.empty((n, 2), dtype=torch.float32, device=x.device)
    check(_lib.load().e2eft_ensemble_minmax(n, npix, _ptr(x), _ptr(out), _ptr(ws), nbytes, _stream()))
    return out


def ensemble_gram(x):
    """fp64 (gram [N, N] = sum_p x_i x_j, sums [N] = sum_p x_i)"""
    n, npix, ws, nbytes = _ens_stack(x)
    gram = torch.empty((n, n), dtype=torch.float64, device=x.device)
    sums = torch.empty(n, dtype=torch.float64, device=x.device)
    check(_lib.load().e2eft_ensemble_gram(n, npix, _ptr(x), _ptr(gram), _ptr(sums), _ptr(ws), nbytes, _stream()))
    return gram, sums


def ensemble_depth_reduce(x, scale, shift, use_mean=False, want_images=True):
    """a_i = x_i * scale_i + shift_i -> (pred, uncertainty, minmax [2]); pred / uncertainty are None when want_images is False"""
    n, npix, ws, nbytes = _ens_stack(x)
    _check_cuda(scale, shift)
    assert scale.dtype == torch.float32 and shift.dtype == torch.float32 and scale.numel() == n and shift.numel() == n
    pred = torch.empty(x.shape[1:], dtype=torch.float32, device=x.device) if want_images else None
    unc = torch.empty(x.shape[1:], dtype=torch.float32, device=x.device) if want_images else None
    minmax = torch.empty(2, dtype=torch.float32, device=x.device)
    check(_lib.load().e2eft_ensemble_depth_reduce(n, npix, _ptr(x), _ptr(scale.contiguous()), _ptr(shift.contiguous()), 1 if use_mean else 0,
                                                  _ptr(pred), _ptr(unc), _ptr(minmax), _ptr(ws), nbytes, _stream()))
    return pred, unc, minmax


def ensemble_depth_finish_(pred, unc, minmax):
    """in place: pred = (pred - min) / (max - min), unc /= (max - min)"""
    _check_cuda(pred, unc, minmax)
    assert pred.dtype == torch.float32 and pred.is_contiguous() and (unc is None or (unc.is_contiguous() and unc.numel() == pred.numel()))
    check(_lib.load().e2eft_ensemble_depth_finish(pred.numel(), _ptr(minmax), _ptr(pred), _ptr(unc), _stream()))
    return pred, unc


def ensemble_normals(x):
    """x [N, 3, H, W] -> (unit vectors [N, 3, H, W], fp64 [N] summed angular error to the mean direction)"""
    n, npix3, ws, nbytes = _ens_stack(x)
    assert x.dim() == 4 and x.shape[1] == 3
    unit = torch.empty_like(x)
    err = torch.empty(n, dtype=torch.float64, device=x.device)
    check(_lib.load().e2eft_ensemble_normals(n, npix3 // 3, _ptr(x), _ptr(unit), _ptr(err), _ptr(ws), nbytes, _stream()))
    return unit, err


# ---------------------------------------------------------------------------------------------------------
# training-sample preparation (csrc/dataprep.hip)
def masked_quantiles(depth, near, far, q_lo=0.02, q_hi=0.98):
    """depth [B, ...] fp32 -> [B, 4] = (q_lo quantile, q_hi quantile, number of valid pixels, ok) over near < depth < far"""
    _check_cuda(depth)
    assert depth.dtype == torch.float32 and depth.is_contiguous() and depth.dim() >= 2
    B = depth.shape[0]
    lib = _lib.load()
    nbytes = lib.e2eft_masked_quantiles_workspace_bytes(B)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=depth.device)
    out = torch.empty((B, 4), dtype=torch.float32, device=depth.device)
    check(lib.e2eft_masked_quantiles(B, depth.numel() // B, _ptr(depth), near, far, q_lo, q_hi, _ptr(out), _ptr(ws), nbytes, _stream()))
    return out


def prepare_sample(rgb01, depth, normal01, near, far, quantiles):
    """rgb01 / normal01 [B,3,H,W] in [0,1], depth [B,1,H,W] metres, quantiles [B,4] -> (rgb, depth3, metric, normals, val_mask bool)"""
    _check_cuda(rgb01, depth, normal01, quantiles)
    B, _, H, W = rgb01.shape
    for t in (rgb01, depth, normal01, quantiles):
        assert t.dtype == torch.float32 and t.is_contiguous()
    assert tuple(depth.shape) == (B, 1, H, W) and tuple(normal01.shape) == (B, 3, H, W) and tuple(quantiles.shape) == (B, 4)
    dev = rgb01.device
    rgb, depth3, normals = torch.empty_like(rgb01), torch.empty_like(rgb01), torch.empty_like(rgb01)
    metric = torch.empty_like(depth)
    mask = torch.empty((B, 1, H, W), dtype=torch.uint8, device=dev)
    check(_lib.load().e2eft_prepare_sample(B, H * W, _ptr(rgb01), _ptr(depth), _ptr(normal01), near, far, _ptr(quantiles), _ptr(rgb), _ptr(depth3),
                                           _ptr(metric), _ptr(normals), _ptr(mask), _stream()))
    return rgb, depth3, metric, normals, mask.bool()


def align_normals_u8(normal_u8, depth, inv_k):
    """Hypersim's camera-facing fix (load.py:185-204,225-232): normal_u8 uint8 [B,H,W,3], depth fp32 [B,H,W] metres, inv_k = 9 floats (row-major inverse
    intrinsics, host) -> uint8 [B,H,W,3]"""
    _check_cuda(normal_u8, depth)
    assert normal_u8.dtype == torch.uint8 and normal_u8.is_contiguous() and normal_u8.dim() == 4 and normal_u8.shape[-1] == 3
    B, H, W, _ = normal_u8.shape
    assert depth.dtype == torch.float32 and depth.is_contiguous() and tuple(depth.shape) == (B, H, W)
    ik = (C.c_double * 9)(*[float(v) for v in inv_k])
    out = torch.empty_like(normal_u8)
    check(_lib.load().e2eft_align_normals_u8(B, H, W, _ptr(normal_u8), _ptr(depth), ik, _ptr(out), _stream()))
    return out


# ------------------------------------------------------------------------------------------------------------
# sample augmentation (csrc/dataaug.hip) and evaluation metrics (csrc/evalmetrics.hip)
def aug_resample_bilinear_u8(img, size, xtab, ytab, flip=None, invert_x_on_flip=False):
    """img uint8 [B,H0,W0,3] -> fp32 [B,3,h,w] in [0,1]: PIL-exact bilinear resize + ToTensor; xtab / ytab = (bounds int32 [out,2], coef int32 [out,k])"""
    _check_cuda(img, xtab[0], xtab[1], ytab[0], ytab[1], flip)
    B, H0, W0, c = img.shape
    assert c == 3 and img.dtype == torch.uint8 and img.is_contiguous()
    h, w = size
    mid = torch.empty((B, H0, w, 3), dtype=torch.uint8, device=img.device)
    out = torch.empty((B, 3, h, w), dtype=torch.float32, device=img.device)
    check(_lib.load().e2eft_aug_resample_bilinear_u8(B, H0, W0, h, w, _ptr(img), _ptr(flip), 1 if invert_x_on_flip else 0, _ptr(xtab[0]), _ptr(xtab[1]),
                                                     xtab[1].shape[1], _ptr(ytab[0]), _ptr(ytab[1]), ytab[1].shape[1], _ptr(mid), _ptr(out), _stream()))
    return out


def aug_gather(img, ymap, xmap, flip=None, invert_x_on_flip=False):
    """index-table gather (nearest resize / crop) with the synchronised flip: fp32 [B,H0,W0] -> [B,h,w], or uint8 [B,H0,W0,3] -> fp32 [B,3,h,w] / 255"""
    _check_cuda(img, ymap, xmap, flip)
    h, w = ymap.numel(), xmap.numel()
    lib = _lib.load()
    if img.dtype == torch.uint8:
        B, H0, W0, c = img.shape
        assert c == 3 and img.is_contiguous()
        out = torch.empty((B, 3, h, w), dtype=torch.float32, device=img.device)
        check(lib.e2eft_aug_gather_u8(B, H0, W0, h, w, _ptr(img), _ptr(ymap), _ptr(xmap), _ptr(flip), 1 if invert_x_on_flip else 0, _ptr(out), _stream()))
        return out
    B, H0, W0 = img.shape
    assert img.dtype == torch.float32 and img.is_contiguous()
    out = torch.empty((B, h, w), dtype=torch.float32, device=img.device)
    check(lib.e2eft_aug_gather_f32(B, H0, W0, h, w, _ptr(img), _ptr(ymap), _ptr(xmap), _ptr(flip), _ptr(out), _stream()))
    return out


def depth_eval(pred, gt, mask, disparity=False, align_max_res=0, min_depth=1e-3, max_depth=80.0, return_aligned=False):
    """pred, gt fp32 [B,H,W], mask bool / uint8 [B,H,W] -> metrics fp32 [B,12] (see include/e2eft.h) [, aligned prediction [B,H,W]]"""
    _check_cuda(pred, gt, mask)
    B, H, W = pred.shape
    pred, gt = pred.float().contiguous(), gt.float().contiguous()
    mask = mask.to(torch.uint8).contiguous()
    lib = _lib.load()
    nws = lib.e2eft_depth_eval_workspace_bytes(B)
    ws = torch.empty(((nws + 7) // 8,), dtype=torch.float64, device=pred.device)
    out = torch.empty((B, 12), dtype=torch.float32, device=pred.device)
    aligned = torch.empty_like(pred) if return_aligned else None
    check(lib.e2eft_depth_eval(B, H, W, _ptr(pred), _ptr(gt), _ptr(mask), 1 if disparity else 0, int(align_max_res or 0), float(min_depth), float(max_depth),
                               _ptr(out), _ptr(aligned), _ptr(ws), nws, _stream()))
    return (out, aligned) if return_aligned else out
