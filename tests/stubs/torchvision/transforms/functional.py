"""torchvision.transforms.functional on TENSORS: resize = torch.nn.functional.interpolate with antialias for bilinear / bicubic
(torchvision 0.17 `_functional_tensor.resize`: float round trip, round + cast back for integer inputs)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import InterpolationMode


def pil_to_tensor(pic):
    img = torch.as_tensor(np.array(pic, copy=True))
    img = img.view(pic.size[1], pic.size[0], len(pic.getbands()))
    return img.permute((2, 0, 1))


def resize(img, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias=True):
    assert isinstance(img, torch.Tensor)
    if isinstance(size, int):
        size = [size]
    if len(size) == 1:
        h, w = img.shape[-2:]
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = size[0], int(size[0] * long / short)
        size = (new_long, new_short) if w <= h else (new_short, new_long)
    mode = interpolation.value if isinstance(interpolation, InterpolationMode) else str(interpolation)
    aa = bool(antialias) and mode in ("bilinear", "bicubic")
    squeeze = img.dim() == 3
    x = img[None] if squeeze else img
    out_dtype = x.dtype
    need_cast = out_dtype not in (torch.float32, torch.float64)
    if need_cast:
        x = x.to(torch.float32)
    kw = {}
    if mode in ("bilinear", "bicubic"):
        kw = {"antialias": aa, "align_corners": False}
    y = F.interpolate(x, size=list(size), mode=mode, **kw)
    if need_cast:
        if mode == "bicubic" and out_dtype == torch.uint8:
            y = y.clamp(0, 255)
        if out_dtype in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
            y = torch.round(y)
        y = y.to(out_dtype)
    return y[0] if squeeze else y
