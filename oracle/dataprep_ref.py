"""TEST INFRASTRUCTURE (see oracle/__init__.py).  CPU restatement of the per-sample arithmetic of the reference's training datasets,
/root/reference/training/dataloaders/load.py:236-283 (Hypersim) and :342-375 (Virtual KITTI 2) — the part after decode + augmentation.
prepare_sample_ref follows the source line by line with the SAME torch calls (torch.quantile, torch.clamp, F.normalize); since round 5 the whole
`__getitem__` of both dataset classes is pinned: tests/golden/make_dataset_golden.py imports load.py (torchvision.transforms / cv2 stood in for by the
few Pillow calls they make) and runs the reference's own Hypersim / VirtualKITTI2 over a synthetic tree; tests/test_datasets_cpu.py checks this file
against that.  MixedDataLoader is pinned the same way (tests/test_data_cpu.py)."""
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


def align_normals_u8_ref(normal_u8, depth_f32):
    """load.py:225-232 with align_normals / creat_uv_mesh (:185-211) on one DECODED sample: normal_u8 uint8 [H,W,3], depth float32 [H,W] metres ->
    the uint8 image the reference hands to its transform.  numpy float64, operation for operation (pinned to the reference's own methods by
    tests/test_datasets_cpu.py, which imports load.py and calls them)."""
    import numpy as np
    H, W = normal_u8.shape[:2]
    n = (normal_u8 / 255.0) * 2.0 - 1.0                                        # :226
    n[:, :, 1:] *= -1                                                          # :228
    K = np.array([[886.81, 0, W / 2], [0, 886.81, H / 2], [0, 0, 1]])          # :230, :194-196
    inv_K = np.linalg.inv(K)
    y, x = np.meshgrid(np.arange(0, H, dtype=np.float64), np.arange(0, W, dtype=np.float64), indexing="ij")
    xy = np.concatenate([np.stack((x, y)).reshape(2, -1), np.ones((1, H * W), dtype=np.float64)], axis=0)      # :206-211
    points = np.matmul(inv_K[:3, :3], xy).reshape(3, H, W)                     # :200
    points = (depth_f32 * points).transpose((1, 2, 0))                         # :201-202
    n[np.sum(n * points, axis=2) > 0] *= -1                                    # :204-205
    n = n * -1                                                                 # :230
    return ((n + 1.0) / 2.0 * 255).astype(np.uint8)                            # :231
