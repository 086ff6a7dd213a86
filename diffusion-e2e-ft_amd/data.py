"""Training input pipeline pieces that belong on the GPU + the dataset mixer.

  MixedDataLoader   /root/reference/training/dataloaders/load.py:18-59 — interleaves two loaders 9:1 (same draw order from
                    numpy's global RNG, same length arithmetic).
  prepare_batch     load.py:236-283 (Hypersim) / :342-375 (Virtual KITTI 2): what __getitem__ does to a decoded + augmented sample,
                    batched on the device: validity mask, 2 % / 98 % depth quantiles, clamp + normalise, normal renormalisation.
  hflip_sample      load.py:76-84: synchronised horizontal flip incl. the sign change of the normals' x component.
  augment_hypersim / augment_vkitti   load.py:67-152: the synchronised transforms (h-flip, PIL-exact bilinear / nearest resize to 480 x 640,
                    ToTensor; h-flip, ToTensor, KITTI benchmark crop 352 x 1216) on batches of DECODED images resident on the device.
File decoding (PNG / EXR through PIL / cv2) stays with the caller: no image library is part of this package."""
import numpy as np
import torch

from . import ops

NEAR_FAR = {"hypersim": (1e-5, 65.0), "vkitti": (1e-5, 80.0)}   # load.py:161, :286
DOMAIN = {"hypersim": "indoor", "vkitti": "outdoor"}            # load.py:283, :375


class MixedDataLoader:
    def __init__(self, loader1, loader2, split1=9, split2=1):
        self.loader1, self.loader2, self.split1, self.split2 = loader1, loader2, split1, split2
        self.frac1, self.frac2 = self.get_split_fractions()
        self.randchoice1 = None

    def get_split_fractions(self):   # load.py:34-41: use all of the scarcer loader, subsample the other to keep split1:split2
        n1, n2 = len(self.loader1), len(self.loader2)
        return min((n2 / n1) * (self.split1 / self.split2), 1), min((n1 / n2) * (self.split2 / self.split1), 1)

    def create_split(self):          # load.py:43-46
        choice = [True] * int(len(self.loader1) * self.frac1) + [False] * int(len(self.loader2) * self.frac2)
        np.random.shuffle(choice)
        return choice

    def __iter__(self):
        self.loader_iter1, self.loader_iter2 = iter(self.loader1), iter(self.loader2)
        self.randchoice1 = self.create_split()
        self.indx = 0
        return self

    def __next__(self):
        if self.indx == len(self.randchoice1):
            raise StopIteration
        first = self.randchoice1[self.indx]
        self.indx += 1
        return next(self.loader_iter1 if first else self.loader_iter2)

    def __len__(self):
        return int(len(self.loader1) * self.frac1) + int(len(self.loader2) * self.frac2)


@ops.tensor_scoped
def hflip_sample(rgb01, depth, normal01=None):
    """synchronised horizontal flip of [..., H, W] tensors; the normal map's x channel (index 0 of dim -3) becomes 1 - x, the float
    form of `255 - x` on the uint8 image (load.py:79-84)."""
    rgb01, depth = rgb01.flip(-1), depth.flip(-1)
    if normal01 is None:
        return rgb01, depth
    n = normal01.flip(-1).clone()
    n[..., 0, :, :] = 1.0 - n[..., 0, :, :]
    return rgb01, depth, n


@torch.no_grad()
@ops.tensor_scoped
def prepare_batch(rgb01, depth, normal01, dataset="hypersim", near_plane=None, far_plane=None):
    """rgb01, normal01: [B,3,H,W] in [0,1] (ToTensor of the decoded images); depth: [B,1,H,W] metres -> the batch dict train.py consumes
    (train.py:470-475): rgb [-1,1], depth [B,3,H,W] in [-1,1], metric [B,1,H,W], normals unit / zero, val_mask bool, domain."""
    near, far = NEAR_FAR[dataset]
    near = near if near_plane is None else near_plane
    far = far if far_plane is None else far_plane
    f = lambda t: t.float().contiguous()
    rgb01, depth, normal01 = f(rgb01), f(depth), f(normal01)
    q = ops.masked_quantiles(depth, near, far, 0.02, 0.98)
    rgb, depth3, metric, normals, mask = ops.prepare_sample(rgb01, depth, normal01, near, far, q)
    return {"rgb": rgb, "depth": depth3, "metric": metric, "normals": normals, "val_mask": mask, "domain": [DOMAIN[dataset]] * rgb.shape[0]}


# ---- synchronised augmentation on the device (csrc/dataaug.hip) ---------------------------------------------------------------------
_PIL_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c) for the bilinear (triangle) filter over the full input
    range: bounds int32 [out, 2] = (first tap, tap count), coefficients int32 [out, ksize] in 22-bit fixed point.  float64 arithmetic in
    Pillow's order, so the integer tables are the ones Pillow builds and the device resize (aug_resample_h/v kernels) is bit-exact."""
    import math
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            v = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - v if v < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w[:xmax] /= ww
        kk[xx] = [int(-0.5 + v * (1 << _PIL_PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PIL_PRECISION_BITS)) for v in w]
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def pil_nearest_map(in_size, out_size):
    """source index of every output index for Image.resize(..., NEAREST): Pillow's ImagingScaleAffine walks xo = scale / 2, xo += scale in
    double precision and truncates — the ACCUMULATED sum, so that exact half-way positions (768 -> 480: (7 + 0.5) * 1.6 = 12) land where
    Pillow's rounding error puts them (11)"""
    a0 = float(in_size) / out_size
    steps = np.full(out_size, a0, dtype=np.float64)
    steps[0] = a0 * 0.5
    xo = np.add.accumulate(steps)          # sequential double additions, as the C loop
    return np.clip(xo.astype(np.int32), 0, in_size - 1)


_TABLES = {}


def _tables(kind, in_size, out_size, device):
    key = (kind, in_size, out_size, str(device))
    if key not in _TABLES:
        if kind == "bilinear":
            b, k = pil_bilinear_coeffs(in_size, out_size)
            _TABLES[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device))
        elif kind == "nearest":
            _TABLES[key] = torch.from_numpy(pil_nearest_map(in_size, out_size)).to(device)
        else:   # ("crop", offset): in_size is the offset here
            _TABLES[key] = (in_size + torch.arange(out_size, dtype=torch.int32)).to(device)
    return _TABLES[key]


def _flip_flags(flip, batch, device):
    if flip is None:
        return None
    return torch.as_tensor(flip, dtype=torch.uint8).reshape(batch).to(device).contiguous()


@torch.no_grad()
@ops.tensor_scoped
def augment_hypersim(rgb_u8, depth, normal_u8=None, size=(480, 640), flip=None):
    """SynchronizedTransform_Hyper (load.py:67-101) on a batch: rgb_u8 / normal_u8 uint8 [B,H0,W0,3] as decoded, depth fp32 [B,H0,W0]; flip:
    per-image booleans (the reference draws `random.random() > 0.5` per sample).  Returns rgb01 [B,3,h,w], depth [B,1,h,w], normal01 or None
    — the inputs of prepare_batch."""
    dev = rgb_u8.device
    B, H0, W0, _ = rgb_u8.shape
    h, w = size
    fl = _flip_flags(flip, B, dev)
    xt, yt = _tables("bilinear", W0, w, dev), _tables("bilinear", H0, h, dev)
    rgb01 = ops.aug_resample_bilinear_u8(rgb_u8.contiguous(), (h, w), xt, yt, flip=fl)
    d = ops.aug_gather(depth.float().contiguous(), _tables("nearest", H0, h, dev), _tables("nearest", W0, w, dev), flip=fl)[:, None]
    n01 = None if normal_u8 is None else ops.aug_resample_bilinear_u8(normal_u8.contiguous(), (h, w), xt, yt, flip=fl, invert_x_on_flip=True)
    return rgb01, d, n01


KB_CROP = (352, 1216)     # KITTI benchmark crop (load.py:112-131)


@torch.no_grad()
@ops.tensor_scoped
def augment_vkitti(rgb_u8, depth, normal_u8=None, flip=None):
    """SynchronizedTransform_VKITTI (load.py:104-152): h-flip, ToTensor, crop the bottom 352 rows / centred 1216 columns"""
    dev = rgb_u8.device
    B, H0, W0, _ = rgb_u8.shape
    ch, cw = KB_CROP
    top, left = int(H0 - ch), int((W0 - cw) / 2)
    fl = _flip_flags(flip, B, dev)
    ym, xm = _tables("crop", top, ch, dev), _tables("crop", left, cw, dev)
    rgb01 = ops.aug_gather(rgb_u8.contiguous(), ym, xm, flip=fl)
    d = ops.aug_gather(depth.float().contiguous(), ym, xm, flip=fl)[:, None]
    n01 = None if normal_u8 is None else ops.aug_gather(normal_u8.contiguous(), ym, xm, flip=fl, invert_x_on_flip=True)
    return rgb01, d, n01
