"""Training input pipeline pieces that belong on the GPU + the dataset mixer.

  MixedDataLoader   /root/reference/training/dataloaders/load.py:18-59 — interleaves two loaders 9:1 (same draw order from
                    numpy's global RNG, same length arithmetic).
  prepare_batch     load.py:236-283 (Hypersim) / :342-375 (Virtual KITTI 2): what __getitem__ does to a decoded + augmented sample,
                    batched on the device: validity mask, 2 % / 98 % depth quantiles, clamp + normalise, normal renormalisation.
  hflip_sample      load.py:76-84: synchronised horizontal flip incl. the sign change of the normals' x component.
File decoding (PNG / EXR through PIL / cv2) and resizing stay with the caller: no image library is part of this package."""
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
