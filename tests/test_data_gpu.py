"""GPU parity of the sample-preparation kernels (csrc/dataprep.hip) against the CPU oracle (oracle/dataprep_ref.py, the reference's
torch calls) and torch.quantile directly."""
import numpy as np
import pytest
import torch

from oracle import dataprep_ref

pytestmark = pytest.mark.gpu


def _depth(n, seed, kind):
    g = torch.Generator().manual_seed(seed)
    if kind == "smooth":
        d = 0.5 + 20 * torch.rand(n, generator=g) ** 2
    elif kind == "ties":                                  # millimetre-quantised depths: long runs of equal values around the quantiles
        d = (torch.randint(400, 9000, (n,), generator=g) // 50 * 50).float() / 1000
    else:                                                 # wide dynamic range incl. values beyond the far plane and invalid zeros
        d = torch.exp(torch.randn(n, generator=g) * 2 + 1)
    d[torch.rand(n, generator=g) < 0.07] = 0.0
    return d


@pytest.mark.parametrize("n", [1, 2, 3, 7, 100, 4099, 480 * 640])
@pytest.mark.parametrize("kind", ["smooth", "ties", "wide"])
def test_masked_quantiles_match_torch(dev, n, kind):
    from diffusion_e2e_ft_amd import ops
    near, far = 1e-5, 65.0
    d = torch.stack([_depth(n, 10 * n + i, kind) for i in range(3)])
    q = ops.masked_quantiles(d.to(dev), near, far).cpu()
    for b in range(3):
        valid = d[b][(d[b] > near) & (d[b] < far)]
        assert int(q[b, 2]) == valid.numel()
        if valid.numel() == 0:
            assert q[b, 3] == 0
            continue
        lo, hi = torch.quantile(valid, 0.02), torch.quantile(valid, 0.98)
        assert abs(float(q[b, 0]) - float(lo)) <= 1e-6 * max(1.0, abs(float(lo))), (float(q[b, 0]), float(lo))
        assert abs(float(q[b, 1]) - float(hi)) <= 1e-6 * max(1.0, abs(float(hi))), (float(q[b, 1]), float(hi))
        assert bool(q[b, 3]) == bool(lo != hi) or abs(float(lo) - float(hi)) < 1e-6
    q2 = ops.masked_quantiles(d.to(dev), near, far).cpu()
    assert torch.equal(q, q2)                              # integer histograms: deterministic


@pytest.mark.parametrize("H,W", [(6, 8), (48, 64), (120, 160)])
def test_prepare_batch_matches_oracle(dev, H, W):
    from diffusion_e2e_ft_amd.data import prepare_batch, NEAR_FAR
    g = torch.Generator().manual_seed(H)
    B = 4
    rgb, nrm = torch.rand(B, 3, H, W, generator=g), torch.rand(B, 3, H, W, generator=g)
    depth = torch.stack([_depth(H * W, 7 + i, k).view(1, H, W) for i, k in enumerate(["smooth", "ties", "wide", "smooth"])])
    depth[3] = 3.0                                         # constant depth: lo == hi -> zeros and an empty mask (load.py:246-249)
    for dataset in ("hypersim", "vkitti"):
        near, far = NEAR_FAR[dataset]
        out = prepare_batch(rgb.to(dev), depth.to(dev), nrm.to(dev), dataset)
        assert out["domain"] == ["indoor" if dataset == "hypersim" else "outdoor"] * B
        for b in range(B):
            ref = dataprep_ref.prepare_sample_ref(rgb[b], depth[b], nrm[b], near, far)
            assert torch.equal(out["rgb"][b].cpu(), ref["rgb"])
            assert torch.equal(out["val_mask"][b].cpu(), ref["val_mask"])
            assert torch.allclose(out["metric"][b].cpu(), ref["metric"], rtol=1e-6, atol=1e-6)
            assert torch.allclose(out["depth"][b].cpu(), ref["depth"], rtol=0, atol=4e-6)
            assert torch.allclose(out["normals"][b].cpu(), ref["normals"], rtol=0, atol=1e-6)
    assert not out["val_mask"][3].any() and (out["depth"][3] == 0).all() and (out["metric"][3] == 0).all()


def test_prepare_batch_nothing_valid(dev):
    from diffusion_e2e_ft_amd.data import prepare_batch
    rgb, nrm = torch.rand(1, 3, 5, 7), torch.rand(1, 3, 5, 7)
    out = prepare_batch(rgb.to(dev), torch.zeros(1, 1, 5, 7, device=dev), nrm.to(dev))
    assert not out["val_mask"].any() and (out["depth"] == 0).all() and (out["metric"] == 0).all() and (out["normals"] == 0).all()
    assert torch.equal(out["rgb"].cpu(), rgb * 2 - 1)


# ---- synchronised augmentation on the device (csrc/dataaug.hip) against Pillow, which is what the reference's transforms call ----------------
def _decoded_batch(B, H0, W0, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, H0), np.linspace(0, 1, W0), indexing="ij")
    rgb = np.stack([np.clip((0.5 + 0.4 * np.sin(6 * xx + b) + 0.1 * rng.standard_normal((H0, W0)))[..., None] * np.array([1.0, 0.8, 0.6]) * 255, 0, 255)
                    for b in range(B)]).astype(np.uint8)
    nrm = rng.integers(0, 256, size=(B, H0, W0, 3), dtype=np.uint8)
    depth = (2.0 + 8.0 * yy[None] + rng.random((B, H0, W0))).astype(np.float32)
    return rgb, depth, nrm


@pytest.mark.parametrize("H0,W0,size", [(96, 128, (60, 80)), (77, 101, (48, 64))])
def test_augment_hypersim_equals_pillow_transforms(dev, H0, W0, size):
    """SynchronizedTransform_Hyper (training/dataloaders/load.py:67-101): flip -> normals' x channel 255 - x -> Resize (PIL bilinear / nearest)
    -> ToTensor, image by image through Pillow, against the batched device kernels: bit-exact"""
    from PIL import Image
    from diffusion_e2e_ft_amd.data import augment_hypersim
    rgb, depth, nrm = _decoded_batch(3, H0, W0, 5)
    flips = [True, False, True]
    h, w = size
    want_rgb, want_d, want_n = [], [], []
    for b in range(3):
        ri, di, ni = Image.fromarray(rgb[b]), Image.fromarray(depth[b], mode="F"), Image.fromarray(nrm[b])
        if flips[b]:
            ri, di, ni = (im.transpose(Image.FLIP_LEFT_RIGHT) for im in (ri, di, ni))
            a = np.array(ni)
            a[:, :, 0] = 255 - a[:, :, 0]
            ni = Image.fromarray(a)
        want_rgb.append(np.asarray(ri.resize((w, h), resample=Image.BILINEAR)).astype(np.float32).transpose(2, 0, 1) / 255.0)
        want_n.append(np.asarray(ni.resize((w, h), resample=Image.BILINEAR)).astype(np.float32).transpose(2, 0, 1) / 255.0)
        want_d.append(np.asarray(di.resize((w, h), resample=Image.NEAREST)))
    r01, d, n01 = augment_hypersim(torch.from_numpy(rgb).to(dev), torch.from_numpy(depth).to(dev), torch.from_numpy(nrm).to(dev), size=size, flip=flips)
    assert tuple(r01.shape) == (3, 3, h, w) and tuple(d.shape) == (3, 1, h, w)
    assert np.array_equal(r01.cpu().numpy(), np.stack(want_rgb)) and np.array_equal(n01.cpu().numpy(), np.stack(want_n))
    assert np.array_equal(d[:, 0].cpu().numpy(), np.stack(want_d))


def test_augment_vkitti_crop_and_flip(dev):
    """SynchronizedTransform_VKITTI (load.py:104-152): flip, ToTensor, KITTI benchmark crop (bottom 352 rows, centred 1216 columns)"""
    from diffusion_e2e_ft_amd.data import augment_vkitti, KB_CROP
    rgb, depth, nrm = _decoded_batch(2, 375, 1242, 7)
    flips = [False, True]
    r01, d, n01 = augment_vkitti(torch.from_numpy(rgb).to(dev), torch.from_numpy(depth).to(dev), torch.from_numpy(nrm).to(dev), flip=flips)
    top, left = 375 - 352, int((1242 - 1216) / 2)
    for b in range(2):
        r, dd, n = rgb[b], depth[b], nrm[b].copy()
        if flips[b]:
            r, dd, n = r[:, ::-1], dd[:, ::-1], n[:, ::-1].copy()
            n[:, :, 0] = 255 - n[:, :, 0]
        sl = (slice(top, top + 352), slice(left, left + 1216))
        assert np.array_equal(r01[b].cpu().numpy(), r[sl].astype(np.float32).transpose(2, 0, 1) / 255.0)
        assert np.array_equal(n01[b].cpu().numpy(), n[sl].astype(np.float32).transpose(2, 0, 1) / 255.0)
        assert np.array_equal(d[b, 0].cpu().numpy(), dd[sl])
    assert tuple(r01.shape[-2:]) == KB_CROP


def test_augmented_batch_feeds_prepare_batch(dev):
    from diffusion_e2e_ft_amd.data import augment_hypersim, prepare_batch
    rgb, depth, nrm = _decoded_batch(2, 96, 128, 9)
    r01, d, n01 = augment_hypersim(torch.from_numpy(rgb).to(dev), torch.from_numpy(depth).to(dev), torch.from_numpy(nrm).to(dev), size=(64, 96), flip=[True, False])
    batch = prepare_batch(r01, d, n01, "hypersim")
    assert tuple(batch["rgb"].shape) == (2, 3, 64, 96) and batch["val_mask"].dtype == torch.bool and torch.isfinite(batch["metric"]).all()
    assert batch["rgb"].min().item() >= -1.0 and batch["rgb"].max().item() <= 1.0
