"""GPU parity of the sample-preparation kernels (csrc/dataprep.hip) against the CPU oracle (oracle/dataprep_ref.py, the reference's
torch calls) and torch.quantile directly."""
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
