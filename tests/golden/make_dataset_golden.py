"""Golden fixture of the reference's training dataset classes (run in THIS container, where /root/reference exists):

    python tests/golden/make_dataset_golden.py        -> tests/golden/dataset_golden.pt

Imports /root/reference/training/dataloaders/load.py and runs ITS `Hypersim` and `VirtualKITTI2` (`_find_pairs`, `__getitem__` with the synchronised
transforms, `align_normals`) over the synthetic trees of tests/dataset_fixture.py.  Two third-party imports of that file are absent from this image and are
stood in for by the handful of Pillow / numpy calls they make on this path (documented below, each with the torchvision / OpenCV behaviour it restates):
`torchvision.transforms.{Resize, RandomHorizontalFlip, ToTensor}` on PIL images and `cv2.imread(path, IMREAD_ANYCOLOR | IMREAD_ANYDEPTH)` of a 16-bit PNG.
pandas, PIL, numpy, torch are the real packages.  The fixture keeps strided samples + float64 sums of every output tensor (tests/dataset_fixture.subsample),
the file lists, the flip coins and the full re-oriented normal image of one sample."""
import importlib.util
import os
import random
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import dataset_fixture as dfx  # noqa: E402

REF = "/root/reference/training/dataloaders/load.py"


def _stubs():
    from PIL import Image
    tv, tr, cv2 = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms"), types.ModuleType("cv2")

    class Resize:                      # torchvision.transforms.Resize on a PIL image: F.resize -> img.resize(size[::-1], interpolation); default BILINEAR
        def __init__(self, size, interpolation=Image.BILINEAR):
            self.size, self.interpolation = size, interpolation

        def __call__(self, img):
            return img.resize((self.size[1], self.size[0]), self.interpolation)

    class RandomHorizontalFlip:        # p = 1.0 in the reference: always F.hflip = img.transpose(FLIP_LEFT_RIGHT) (it also draws torch.rand(1), kept)
        def __init__(self, p=0.5):
            self.p = p

        def __call__(self, img):
            return img.transpose(Image.FLIP_LEFT_RIGHT) if torch.rand(1) < self.p else img

    class ToTensor:                    # F.to_tensor: uint8 HWC -> float CHW / 255; mode "F" -> [1,H,W] unscaled
        def __call__(self, pic):
            if pic.mode == "F":
                return torch.from_numpy(np.array(pic, np.float32, copy=True))[None]
            a = torch.from_numpy(np.array(pic, np.uint8, copy=True))
            return a.permute(2, 0, 1).contiguous().to(torch.float32).div(255)

    tr.Resize, tr.RandomHorizontalFlip, tr.ToTensor = Resize, RandomHorizontalFlip, ToTensor
    tv.transforms = tr
    cv2.IMREAD_ANYCOLOR, cv2.IMREAD_ANYDEPTH = 4, 2
    cv2.imread = lambda path, flags=None: np.array(Image.open(path))       # a single-channel 16-bit PNG: the stored integers, as OpenCV returns them
    return {"torchvision": tv, "torchvision.transforms": tr, "cv2": cv2}


def import_reference_load():
    stubs = _stubs()
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        spec = importlib.util.spec_from_file_location("ref_load_full", REF)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return m


def run_reference(tmp):
    """-> golden dict; also used by tests/test_datasets_cpu.py to re-derive it when the reference is present"""
    ref = import_reference_load()
    out = {}
    root_dir, split_path = dfx.make_hypersim_tree(tmp)
    vroot = dfx.make_vkitti_tree(tmp)
    cwd = os.getcwd()
    work = os.path.join(tmp, "cwd")
    os.makedirs(os.path.join(work, "data/hypersim/processed/train"), exist_ok=True)
    shutil.copy(split_path, os.path.join(work, "data/hypersim/processed/train/filename_meta_train.csv"))      # the reference hard-codes this relative path (load.py:163)
    os.chdir(work)
    try:
        hs = ref.Hypersim(root_dir=root_dir, transform=True)
    finally:
        os.chdir(cwd)
    vk = ref.VirtualKITTI2(root_dir=vroot, transform=True)
    vk.pairs.sort()              # the reference keeps os.listdir's order (file-system dependent); the product sorts: same index -> file map on every box
    rel = lambda p, base: os.path.relpath(p, base)
    out["hypersim_pairs"] = [{k: rel(v, root_dir) for k, v in pr.items()} for pr in hs.pairs]
    out["vkitti_pairs"] = sorted(tuple(rel(p, vroot) for p in pr) for pr in vk.pairs)
    for name, ds, n in (("hypersim", hs, len(hs)), ("vkitti", vk, len(vk))):
        for seed in range(1000):      # a coin sequence that shows both faces
            random.seed(seed)
            coins = [random.random() > 0.5 for _ in range(n)]
            if len(set(coins)) == 2 or n < 2:
                break
        random.seed(seed)
        samples = [ds[i] for i in range(n)]
        out[name] = {"seed": seed, "coins": coins, "len": n, "domain": [s["domain"] for s in samples],
                     "samples": [{k: dfx.subsample(s[k]) for k in ("rgb", "depth", "metric", "normals", "val_mask")} for s in samples]}
    # the orientation fix alone, through the reference's own methods (load.py:225-232)
    from PIL import Image
    pr = hs.pairs[0]
    nimg = Image.open(pr["normal_path"]).convert("RGB")
    dimg = Image.fromarray(np.array(Image.open(pr["depth_path"])) / 1000)
    na = (np.array(nimg) / 255.0) * 2.0 - 1.0
    H, W = na.shape[:2]
    na[:, :, 1:] *= -1
    na = hs.align_normals(na, np.array(dimg), [886.81, 886.81, W / 2, H / 2], H, W) * -1
    out["aligned_normal_u8_sample0"] = torch.from_numpy(((na + 1.0) / 2.0 * 255).astype(np.uint8))
    # untransformed branch (transform=None -> plain ToTensor): one sample each
    os.chdir(work)
    try:
        hs0 = ref.Hypersim(root_dir=root_dir, transform=False)
    finally:
        os.chdir(cwd)
    s = hs0[1]
    out["hypersim_untransformed_sample1"] = {k: dfx.subsample(s[k]) for k in ("rgb", "depth", "metric", "normals", "val_mask")}
    return out


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as tmp:
        g = run_reference(tmp)
    torch.save(g, os.path.join(HERE, "dataset_golden.pt"))
    print("hypersim", g["hypersim"]["len"], g["hypersim"]["coins"], "vkitti", g["vkitti"]["len"], g["vkitti"]["coins"],
          "bytes", os.path.getsize(os.path.join(HERE, "dataset_golden.pt")))
