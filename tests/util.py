"""Shared helpers for the parity tests."""
import torch

DTYPES = [torch.float32, torch.float16, torch.bfloat16]
# max |out - ref| / max |ref| tolerances for a single op with fp32 accumulation and output rounding to dtype
TOL = {torch.float32: 3e-5, torch.float16: 3e-3, torch.bfloat16: 2e-2}


def rel_err(out, ref):
    out = out.detach().double().cpu()
    ref = ref.detach().double().cpu()
    denom = ref.abs().max().clamp_min(1e-30)
    return ((out - ref).abs().max() / denom).item()


def assert_close(out, ref, dtype, what="", scale=1.0):
    assert out.shape == ref.shape, (what, out.shape, ref.shape)
    assert torch.isfinite(out.float()).all(), "%s: non-finite output" % what
    e = rel_err(out, ref)
    assert e <= TOL[dtype] * scale, "%s: rel err %.3e > %.3e (%s)" % (what, e, TOL[dtype] * scale, dtype)
    return e


def q(t, dtype):
    """quantize a fp32 CPU tensor to dtype and back (so reference and kernel see identical inputs)"""
    return t.to(dtype).float()


def nhwc(t, dtype, dev):
    """NCHW fp32 cpu -> NHWC dtype device"""
    return t.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)


def to_nchw(t):
    return t.permute(0, 3, 1, 2).float().cpu()


def pack_conv_weight(w, dtype, dev, cin_pad=None):
    """[Co,Ci,kh,kw] -> [Co, kh*kw*Ci(_pad)] OHWI"""
    Co, Ci, kh, kw = w.shape
    w = w.permute(0, 2, 3, 1)
    if cin_pad is not None and cin_pad != Ci:
        w = torch.nn.functional.pad(w, (0, cin_pad - Ci))
    return w.reshape(Co, -1).contiguous().to(dtype).to(dev)
