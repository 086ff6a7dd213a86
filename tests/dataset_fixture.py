"""Deterministic synthetic Hypersim / Virtual KITTI 2 trees in the reference's on-disk layout (training/dataloaders/load.py:163,170-183,294-318): PNG / JPEG-named
files written with Pillow from seeded numpy arrays, so that the CPU container (where the reference's dataset classes are run from source to make the golden
fixture) and the GPU box (where the product's DeviceLoader is checked against it) read byte-identical files.  TEST INFRASTRUCTURE."""
import csv
import os

import numpy as np


def _scene(rng, H, W):
    """rgb uint8, depth in metres (float64), normals uint8 of a tilted plane with boxes: smooth regions + edges + some invalid (0 / far) depth"""
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    c = rng.random(6)
    depth = 4.0 + 3.0 * (c[0] - 0.5) * xx + 3.0 * (c[1] - 0.5) * yy + 6.0 * c[2]
    for _ in range(3):
        r = rng.random(5)
        x0, y0 = r[0] * 1.4 - 1.0, r[1] * 1.4 - 1.0
        box = (xx > x0) & (xx < x0 + 0.2 + 0.5 * r[2]) & (yy > y0) & (yy < y0 + 0.2 + 0.5 * r[3])
        depth = np.where(box, 0.6 + 5.0 * r[4], depth)
    depth[:2, :] = 0.0                                   # invalid rows (below the near plane)
    depth[-1, : W // 3] = 200.0                          # beyond the far plane (uint16 millimetres would overflow: callers clip)
    rgb = np.clip(128 + 90 * np.sin(6 * xx * c[3] + 3 * yy)[..., None] * np.array([1.0, 0.6, -0.8]) + rng.normal(0, 12, (H, W, 3)), 0, 255).astype(np.uint8)
    n = np.stack([0.4 * np.sin(5 * xx + c[4]), 0.4 * np.cos(4 * yy + c[5]), np.ones_like(xx)], axis=-1) + rng.normal(0, 0.05, (H, W, 3))
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    flipm = rng.random((H, W)) > 0.7                      # some normals stored facing away from the camera (what align_normals repairs)
    n[flipm] *= -1
    normal = np.clip((n + 1.0) / 2.0 * 255.0 + 0.5, 0, 255).astype(np.uint8)
    return rgb, depth, normal


def make_hypersim_tree(root, n=3, H=96, W=128, seed=7):
    """-> (root_dir, split_path): `n` train samples + one row that is not in the public release + one of another split + one with a missing normal map"""
    from PIL import Image
    rng = np.random.default_rng(seed)
    root_dir = os.path.join(root, "hypersim", "processed")
    split_path = os.path.join(root, "hypersim", "filename_meta_train.csv")
    rows = []
    for i in range(n + 3):
        scene, cam, frame = "ai_%03d_001" % (i + 1), "cam_00", i * 3
        rgb_rel = os.path.join(scene, "rgb_%s_fr%04d.png" % (cam, frame))
        depth_rel = os.path.join(scene, "depth_plane_%s_fr%04d.png" % (cam, frame))
        rgb, depth, normal = _scene(rng, H, W)
        os.makedirs(os.path.join(root_dir, "train", scene), exist_ok=True)
        Image.fromarray(rgb).save(os.path.join(root_dir, "train", rgb_rel), compress_level=1)
        Image.fromarray(np.clip(depth * 1000.0, 0, 65535).astype(np.uint16)).save(os.path.join(root_dir, "train", depth_rel), compress_level=1)
        ndir = os.path.join(root_dir, "normals", scene, "images", "scene_%s_geometry_preview" % cam)
        os.makedirs(ndir, exist_ok=True)
        if i != n + 2:                                    # the last row has no normal map on disk: skipped by _find_pairs
            Image.fromarray(normal).save(os.path.join(ndir, "frame.%04d.normal_cam.png" % frame), compress_level=1)
        rows.append({"included_in_public_release": "False" if i == n else "True", "split_partition_name": "val" if i == n + 1 else "train",
                     "rgb_path": rgb_rel, "depth_path": depth_rel, "scene_name": scene, "camera_name": cam, "frame_id": frame})
    os.makedirs(os.path.dirname(split_path), exist_ok=True)
    with open(split_path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    return root_dir, split_path


def make_vkitti_tree(root, n=2, H=375, W=1242, seed=11):
    """-> root_dir: `n` frames of Scene01/morning/Camera_0 (the reference lists `rgb_*.jpg`; PNG bytes behind that name decode the same through Pillow — a
    lossless container keeps the golden values independent of the JPEG codec build)"""
    from PIL import Image
    rng = np.random.default_rng(seed)
    root_dir = os.path.join(root, "vkitti")
    dirs = [os.path.join(root_dir, d, "Scene01", "morning", "frames", k, "Camera_0") for d, k in
            (("vkitti_2.0.3_rgb", "rgb"), ("vkitti_2.0.3_depth", "depth"), ("vkitti_DAG_normals", "normal"))]
    for d in dirs:
        os.makedirs(d, exist_ok=True)
    for i in range(n):
        rgb, depth, normal = _scene(rng, H, W)
        depth = depth * 6.0                                # outdoor range; the far rows exceed 80 m
        Image.fromarray(rgb).save(os.path.join(dirs[0], "rgb_%05d.jpg" % i), format="PNG", compress_level=1)
        Image.fromarray(np.clip(depth * 100.0, 0, 65535).astype(np.uint16)).save(os.path.join(dirs[1], "depth_%05d.png" % i), compress_level=1)
        Image.fromarray(normal).save(os.path.join(dirs[2], "normal_%05d.png" % i), compress_level=1)
    return root_dir


def subsample(t):
    """what the golden fixture keeps of a [C,H,W] tensor: a strided sample + float64 sums (the full tensors would be tens of MB)"""
    import torch
    t = t.detach().cpu()
    x = t.double() if t.dtype != torch.bool else t.double()
    return {"shape": tuple(t.shape), "sample": t[..., ::7, ::11].clone(), "sum": float(x.sum()), "abs_sum": float(x.abs().sum())}
