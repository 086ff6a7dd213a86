"""Generate tests/golden/eval_golden.pt: outputs of the REFERENCE'S evaluation functions (Marigold/src/util/alignment.py, metric.py, imported
from /root/reference) on seeded synthetic (prediction, ground truth, mask) triples, chained as Marigold/eval.py:172-209 chains them.
Run from the repo root: `python tests/golden/make_eval_golden.py`."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/Marigold/src/util"
HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ("abs_relative_difference", "squared_relative_difference", "rmse_linear", "rmse_log", "log10", "delta1_acc", "delta2_acc", "delta3_acc",
         "i_rmse", "silog_rmse")


def _import(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def eval_case(seed, H, W, invalid=0.1):
    """metric depth = plane + boxes in [0.5, 20] m; prediction = affine-distorted, noisy copy in [0, 1] (what the pipeline returns)"""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    gt = 2.0 + 6.0 * yy + 3.0 * xx * yy + 2.0 * (torch.rand(1, generator=g) > 0.5).float() * ((xx > 0.3) & (xx < 0.6)).float()
    gt = gt + 0.05 * torch.randn(H, W, generator=g)
    pred = ((gt - gt.min()) / (gt.max() - gt.min()) * 0.8 + 0.1 + 0.02 * torch.randn(H, W, generator=g)).clamp(0.0, 1.0)
    mask = torch.rand(H, W, generator=g) > invalid
    return pred.float(), gt.float().clamp(0.5, 20.0), mask


CASES = [dict(seed=1, H=48, W=64), dict(seed=2, H=37, W=53), dict(seed=3, H=96, W=128, invalid=0.5), dict(seed=4, H=60, W=80)]
SETTINGS = [dict(alignment="least_square", max_res=None), dict(alignment="least_square", max_res=32), dict(alignment="least_square_disparity", max_res=None)]


def reference_eval(al, met, pred, gt, mask, alignment, max_res, dmin, dmax):
    p, g, m = pred.numpy(), gt.numpy(), mask.numpy()
    if alignment == "least_square":
        a, scale, shift = al.align_depth_least_square(gt_arr=g, pred_arr=p, valid_mask_arr=m, return_scale_shift=True, max_resolution=max_res)
    else:
        gd, gmask = al.depth2disparity(depth=g, return_mask=True)
        vm = m & gmask & (p > 0)
        dp, scale, shift = al.align_depth_least_square(gt_arr=gd, pred_arr=p, valid_mask_arr=vm, return_scale_shift=True, max_resolution=max_res)
        dp = np.clip(dp, a_min=1e-3, a_max=None)
        a = al.disparity2depth(dp)
    a = np.clip(a, a_min=dmin, a_max=dmax)
    a = np.clip(a, a_min=1e-6, a_max=None)
    at = torch.from_numpy(a)
    vals = [float(getattr(met, n)(at.clone(), gt.clone(), mask.clone())) for n in NAMES]
    return torch.tensor(vals, dtype=torch.float64), float(np.asarray(scale).reshape(-1)[0]), float(np.asarray(shift).reshape(-1)[0]), at


def main():
    if "pandas" not in sys.modules:
        try:
            import pandas  # noqa: F401
        except Exception:
            sys.modules["pandas"] = types.ModuleType("pandas")
    al = _import(os.path.join(REF, "alignment.py"), "ref_alignment")
    met = _import(os.path.join(REF, "metric.py"), "ref_metric")
    out = {}
    for ci, c in enumerate(CASES):
        pred, gt, mask = eval_case(**c)
        for si, s in enumerate(SETTINGS):
            vals, scale, shift, aligned = reference_eval(al, met, pred, gt, mask, s["alignment"], s["max_res"], 1e-3, 80.0)
            out[(ci, si)] = {"metrics": vals, "scale": scale, "shift": shift, "aligned_sample": aligned[::7, ::5].clone()}
            print(ci, s, ["%.5f" % v for v in vals.tolist()], "scale %.4f shift %.4f" % (scale, shift))
    torch.save(out, os.path.join(HERE, "eval_golden.pt"))


if __name__ == "__main__":
    main()
