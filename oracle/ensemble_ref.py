"""TEST INFRASTRUCTURE (see oracle/__init__.py).  CPU restatement of the reference's test-time ensembling:

  ensemble_depths_ref   /root/reference/Marigold/marigold/util/ensemble.py:40-132
  ensemble_normals_ref  /root/reference/Marigold/marigold/marigold_pipeline.py:58-71

PINNED: tests/golden/ensemble_golden.pt holds the outputs of the reference's own functions (imported from /root/reference by
tests/golden/make_ensemble_golden.py) on seeded stacks; tests/test_oracle_pins.py checks this restatement against them, and
against the live reference when that tree is present."""
import numpy as np
import torch
from scipy.optimize import minimize


def _objective(stack, s, t, reduction, strength):
    """ensemble.py:77-101: RMS of all pairwise differences of the aligned stack + range regulariser on its per-pixel reduction"""
    aligned = stack * s.view(-1, 1, 1) + t.view(-1, 1, 1)
    i, j = torch.triu_indices(stack.shape[0], stack.shape[0], offset=1)
    rms = torch.sqrt(torch.mean((aligned[i] - aligned[j]) ** 2))
    centre = aligned.mean(0) if reduction == "mean" else aligned.median(0).values
    return rms + ((0 - centre.min()).abs() + (1 - centre.max()).abs()) * strength


def ensemble_depths_ref(stack, regularizer_strength=0.02, max_iter=2, tol=1e-3, reduction="median", max_res=None):
    if reduction not in ("median", "mean"):
        raise ValueError(reduction)
    full = stack.clone()
    n = stack.shape[0]
    work = stack
    if max_res is not None:                                   # :61-65 (nn.Upsample on a 3-D tensor: last axis only)
        f = torch.min(max_res / torch.tensor(stack.shape[-2:]))
        if f < 1:
            work = torch.nn.Upsample(scale_factor=f, mode="nearest")(stack)
    flat = work.reshape(n, -1).numpy()
    lo, hi = flat.min(1), flat.max(1)                         # :68-72
    s0 = 1.0 / (hi - lo)
    x0 = np.concatenate([s0, -s0 * lo]).astype(np.float32)

    def closure(x):
        s = torch.from_numpy(np.asarray(x[:n])).to(work.dtype)
        t = torch.from_numpy(np.asarray(x[n:])).to(work.dtype)
        return _objective(work, s, t, reduction, regularizer_strength).numpy().astype(np.float32)

    x = minimize(closure, x0, method="BFGS", tol=tol, options={"maxiter": max_iter, "disp": False}).x
    s = torch.from_numpy(np.asarray(x[:n])).to(full.dtype)
    t = torch.from_numpy(np.asarray(x[n:])).to(full.dtype)
    aligned = full * s.view(-1, 1, 1) + t.view(-1, 1, 1)      # :113-127
    if reduction == "mean":
        centre, spread = aligned.mean(0), aligned.std(0)
    else:
        centre = aligned.median(0).values
        spread = (aligned - centre).abs().median(0).values
    lo, hi = centre.min(), centre.max()                       # :129-133
    return (centre - lo) / (hi - lo), spread / (hi - lo)


def ensemble_normals_ref(stack):
    """stack [N,3,H,W] -> (unit normals of the member closest to the mean direction [3,H,W], None)"""
    unit = stack / (stack.norm(p=2, dim=1, keepdim=True) + 1e-5)
    phi = torch.atan2(unit[:, 1], unit[:, 0]).mean(0)
    theta = torch.atan2(unit[:, :2].norm(p=2, dim=1), unit[:, 2]).mean(0)
    mean_dir = torch.stack([theta.sin() * phi.cos(), theta.sin() * phi.sin(), theta.cos()])
    ang = torch.acos(torch.clip(torch.cosine_similarity(mean_dir[None], unit, dim=1), -0.999, 0.999))
    return unit[int(torch.argmin(ang.reshape(stack.shape[0], -1).sum(-1)))], None


def normals_error_sums_ref(stack):
    """the per-member summed angular error that drives the argmin above (fp64 sums for the kernel test)"""
    unit = stack / (stack.norm(p=2, dim=1, keepdim=True) + 1e-5)
    phi = torch.atan2(unit[:, 1], unit[:, 0]).mean(0)
    theta = torch.atan2(unit[:, :2].norm(p=2, dim=1), unit[:, 2]).mean(0)
    mean_dir = torch.stack([theta.sin() * phi.cos(), theta.sin() * phi.sin(), theta.cos()])
    ang = torch.acos(torch.clip(torch.cosine_similarity(mean_dir[None], unit, dim=1), -0.999, 0.999))
    return unit, ang.double().reshape(stack.shape[0], -1).sum(-1)
