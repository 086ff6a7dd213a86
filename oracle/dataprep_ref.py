"""TEST INFRASTRUCTURE (see oracle/__init__.py).  CPU restatement of the per-sample arithmetic of the reference's training datasets,
/root/reference/training/dataloaders/load.py:236-283 (Hypersim) and :342-375 (Virtual KITTI 2) — the part after decode + augmentation.
PARITY UNPINNED at the file level: load.py cannot be imported here (torchvision, pandas, cv2, PIL are absent), so this follows the
source line by line with the SAME torch calls (torch.quantile, torch.clamp, F.normalize).  MixedDataLoader IS pinned: the test
imports load.py with those four imports stubbed and runs the reference's own class."""
import torch


def prepare_sample_ref(rgb01, depth, normal01, near, far):
    """one sample: rgb01 / normal01 [3,H,W] in [0,1], depth [1,H,W] -> dict as load.py:283"""
    valid = (depth > near) & (depth < far)                                     # :236
    rgb = rgb01 * 2.0 - 1.0                                                    # :239
    if valid.any():                                                            # :242
        flat = depth[valid].flatten().float()
        lo, hi = torch.quantile(flat, 0.02), torch.quantile(flat, 0.98)
        if lo == hi:                                                           # :246
            d = torch.zeros_like(depth)
            metric = torch.zeros_like(depth)
            valid = torch.zeros_like(depth).bool()
        else:
            d = torch.clamp(depth, lo, hi)                                     # :250
            d[~valid] = hi
            metric = d.clone()
            d = torch.clamp(((d - lo) / (hi - lo)) * 2.0 - 1.0, -1, 1)         # :253
    else:
        d = torch.zeros_like(depth)
        metric = torch.zeros_like(depth)
    depth3 = torch.stack([d, d, d]).squeeze()                                  # :257
    n = torch.nn.functional.normalize(normal01 * 2.0 - 1.0, p=2, dim=0)        # :260-261
    n = n.clone()
    n[:, ~valid.squeeze()] = 0                                                 # :263-265
    return {"rgb": rgb, "depth": depth3, "metric": metric, "normals": n, "val_mask": valid}
