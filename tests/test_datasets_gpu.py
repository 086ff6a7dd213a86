"""Training input pipeline on the GPU: data.Hypersim / data.VirtualKITTI2 + data.DeviceLoader against tests/golden/dataset_golden.pt — the
outputs of the REFERENCE's own dataset classes (training/dataloaders/load.py:160-375, run from source in the CPU container by
tests/golden/make_dataset_golden.py over the same synthetic file tree tests/dataset_fixture.py writes here).  rgb and the validity mask must match bit
for bit (uint8 -> float is exact; the resize is Pillow's integer arithmetic), depth / metric / normals to 4e-6 (float32 quantile interpolation order)."""
import os
import random
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import dataset_fixture as dfx  # noqa: E402
from oracle import dataprep_ref  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = torch.load(os.path.join(HERE, "golden", "dataset_golden.pt"))
KEYS = ("rgb", "depth", "metric", "normals", "val_mask")


def _check(batch, i, gold, exact=("rgb", "val_mask"), tol=4e-6):
    for k in KEYS:
        got, want = dfx.subsample(batch[k][i]), gold[k]
        assert got["shape"] == want["shape"], (k, got["shape"], want["shape"])
        if k == "val_mask":
            assert torch.equal(got["sample"], want["sample"]) and got["sum"] == want["sum"], k
        else:
            err = (got["sample"].double() - want["sample"].double()).abs().max().item()
            assert err <= (0.0 if k in exact else tol), (k, err)
            assert abs(got["sum"] - want["sum"]) <= 2e-6 * max(1.0, want["abs_sum"]), (k, got["sum"], want["sum"])


@pytest.fixture(scope="module")
def trees(tmp_path_factory):
    from diffusion_e2e_ft_amd import data
    tmp = str(tmp_path_factory.mktemp("datasets"))
    root_dir, split_path = dfx.make_hypersim_tree(tmp)
    vroot = dfx.make_vkitti_tree(tmp)
    return data.Hypersim(root_dir, transform=True, split_path=split_path), data.VirtualKITTI2(vroot, transform=True), root_dir, split_path, vroot


@pytest.mark.parametrize("name", ["hypersim", "vkitti"])
def test_device_loader_reproduces_the_reference_datasets(dev, trees, name):
    from diffusion_e2e_ft_amd.data import DeviceLoader
    ds = trees[0] if name == "hypersim" else trees[1]
    g = GOLD[name]
    assert len(ds) == g["len"]
    random.seed(g["seed"])            # the reference draws its flip coin from Python's `random`, once per sample, in sample order
    loader = DeviceLoader(ds, batch_size=len(ds), device=dev, shuffle=False, workers=2)
    batches = list(loader)
    assert len(batches) == 1 == len(loader)
    b = batches[0]
    assert b["domain"] == g["domain"] and b["rgb"].device.type == "cuda" and b["val_mask"].dtype == torch.bool
    torch.cuda.synchronize()
    for i in range(len(ds)):
        _check(b, i, g["samples"][i])
    # batch size 1, two workers, prefetch 2: same samples one by one (coins re-drawn in the same order), every batch usable on the consumer's stream
    random.seed(g["seed"])
    for i, b1 in enumerate(DeviceLoader(ds, batch_size=1, device=dev, shuffle=False, workers=2, prefetch=2)):
        _check(b1, 0, g["samples"][i])
    loader.close()


def test_orientation_fix_kernel_is_bit_exact(dev, trees):
    from diffusion_e2e_ft_amd import ops
    from diffusion_e2e_ft_amd.data import Hypersim
    hs = trees[0]
    s = hs[0]
    n = torch.from_numpy(s["normal_u8"])[None].to(dev)
    d = torch.from_numpy(s["depth"])[None].to(dev)
    got = ops.align_normals_u8(n, d, Hypersim.inverse_intrinsics(96, 128).reshape(-1))[0].cpu()
    assert torch.equal(got, GOLD["aligned_normal_u8_sample0"])              # the reference's own align_normals, through the fixture
    # full Hypersim resolution, random content, a batch: against the numpy restatement (pinned to the reference in tests/test_datasets_cpu.py)
    rng = np.random.default_rng(3)
    B, H, W = 2, 768, 1024
    nn = rng.integers(0, 256, size=(B, H, W, 3), dtype=np.uint8)
    dd = (rng.random((B, H, W)) * 30).astype(np.float32)
    dd[0, :5] = 0
    got = ops.align_normals_u8(torch.from_numpy(nn).to(dev), torch.from_numpy(dd).to(dev), Hypersim.inverse_intrinsics(H, W).reshape(-1)).cpu().numpy()
    for b in range(B):
        assert np.array_equal(got[b], dataprep_ref.align_normals_u8_ref(nn[b], dd[b])), b


def test_untransformed_branch_and_mixed_loader(dev, trees):
    from diffusion_e2e_ft_amd import data
    hs, vk, root_dir, split_path, vroot = trees
    hs0 = data.Hypersim(root_dir, transform=False, split_path=split_path)
    b = next(iter(data.DeviceLoader(hs0, batch_size=3, device=dev, shuffle=False, workers=1)))
    torch.cuda.synchronize()
    _check(b, 1, GOLD["hypersim_untransformed_sample1"])
    assert tuple(b["rgb"].shape) == (3, 3, 96, 128)
    # the reference's training input: MixedDataLoader(hypersim loader, vkitti loader, 9, 1) (train.py:364-366) over the device loaders
    np.random.seed(0)
    mixed = data.MixedDataLoader(data.DeviceLoader(hs, batch_size=1, device=dev, workers=2), data.DeviceLoader(vk, batch_size=1, device=dev, workers=2), 9, 1)
    seen = [bb["domain"][0] + str(tuple(bb["rgb"].shape[-2:])) for bb in mixed]
    assert len(seen) == len(mixed) and set(seen) <= {"indoor(480, 640)", "outdoor(352, 1216)"} and "indoor(480, 640)" in seen


def test_loader_feeds_a_training_micro_step(dev, trees):
    """the batch dict is what training.e2e_ft_loss consumes (train.py:470-475): one micro-step on a loader batch (tiny UNet / VAE), finite loss and gradients"""
    import golden_cases as gc
    from oracle import config
    from diffusion_e2e_ft_amd import data, training
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    from diffusion_e2e_ft_amd.vae import AutoencoderKL
    unet = UNet2DConditionModel(**config.TINY_UNET)
    unet.load_state_dict(gc.tiny_unet_sd())
    vae = AutoencoderKL(**config.TINY_VAE)
    vae.load_state_dict(gc.tiny_vae_sd())
    unet, vae = unet.to(dev).train(), vae.to(dev).eval().requires_grad_(False)
    _, text = gc.train_batch()
    b = next(iter(data.DeviceLoader(trees[0], batch_size=2, device=dev, shuffle=False, workers=2)))
    loss = training.e2e_ft_loss(unet, vae, b, text, "depth")
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in unet.parameters() if p.grad is not None)
