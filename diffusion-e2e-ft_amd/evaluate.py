"""Depth evaluation on the device: the arithmetic of the reference's evaluation loop — least-squares alignment
(Marigold/src/util/alignment.py:8-56), range clipping (Marigold/eval.py:203-209) and the metric suite (Marigold/src/util/metric.py:34-158) —
for whole batches in one launch sequence (csrc/evalmetrics.hip), instead of numpy.linalg.lstsq on the host plus ten torch reductions per image.

  align_depth_least_square   same name and return convention as the reference's function, torch tensors on the device
  depth_metrics              per-image metric table (dict name -> tensor [B]) and its mean, named as eval.py's metric functions
  MetricTracker              running averages keyed by metric name (the reference's pandas-backed tracker, metric.py:9-31)
"""
import torch

from . import ops

METRIC_NAMES = ("abs_relative_difference", "squared_relative_difference", "rmse_linear", "rmse_log", "log10", "delta1_acc", "delta2_acc",
                "delta3_acc", "i_rmse", "silog_rmse")


def _b(t):
    """[H,W] / [B,H,W] / [B,1,H,W] -> [B,H,W]; anything else is an error (a [B,3,H,W] prediction must not spin here)"""
    t = torch.as_tensor(t)
    if t.dim() == 4:
        if t.shape[1] != 1:
            raise ValueError("expected a depth map [B,1,H,W], [B,H,W] or [H,W]; got shape %s" % (tuple(t.shape),))
        t = t[:, 0]
    if t.dim() == 2:
        return t[None]
    if t.dim() != 3:
        raise ValueError("expected a depth map [B,1,H,W], [B,H,W] or [H,W]; got shape %s" % (tuple(t.shape),))
    return t


@torch.no_grad()
@ops.tensor_scoped
def align_depth_least_square(gt_arr, pred_arr, valid_mask_arr, return_scale_shift=True, max_resolution=None):
    """alignment.py:8-56: returns pred * scale + shift (no clipping) [, scale, shift]; inputs [H,W] or [B,H,W] device tensors"""
    gt, pred, mask = _b(gt_arr).float(), _b(pred_arr).float(), _b(valid_mask_arr)
    m = ops.depth_eval(pred, gt, mask, align_max_res=max_resolution or 0, min_depth=1e-30, max_depth=3e38)
    scale, shift = m[:, 10], m[:, 11]
    aligned = (pred * scale[:, None, None] + shift[:, None, None]).reshape(torch.as_tensor(pred_arr).shape)
    if return_scale_shift:
        return (aligned, scale, shift) if scale.numel() > 1 else (aligned, scale[0], shift[0])
    return aligned


@torch.no_grad()
@ops.tensor_scoped
def depth_metrics(pred, gt, valid_mask, alignment="least_square", min_depth=1e-3, max_depth=80.0, alignment_max_res=None, return_aligned=False):
    """pred (affine-invariant prediction), gt (metric depth), valid_mask: [B,H,W] (or [H,W]) device tensors.  alignment: "least_square" |
    "least_square_disparity" (eval.py:172-201).  Returns {metric name: tensor [B]} plus "scale" / "shift" (and "aligned" when asked)."""
    if alignment not in ("least_square", "least_square_disparity"):
        raise ValueError("alignment must be least_square or least_square_disparity (the prediction is affine-invariant)")
    res = ops.depth_eval(_b(pred).float(), _b(gt).float(), _b(valid_mask), disparity=alignment == "least_square_disparity",
                         align_max_res=alignment_max_res or 0, min_depth=min_depth, max_depth=max_depth, return_aligned=return_aligned)
    table, aligned = res if return_aligned else (res, None)
    out = {n: table[:, i] for i, n in enumerate(METRIC_NAMES)}
    out["scale"], out["shift"] = table[:, 10], table[:, 11]
    if return_aligned:
        out["aligned"] = aligned
    return out


class MetricTracker:
    """metric.py:9-31 without pandas: update(key, value, n) / avg(key) / result()"""

    def __init__(self, *keys):
        self._tot = {k: 0.0 for k in keys}
        self._cnt = {k: 0 for k in keys}

    def reset(self):
        for k in self._tot:
            self._tot[k], self._cnt[k] = 0.0, 0

    def update(self, key, value, n=1):
        self._tot[key] = self._tot.get(key, 0.0) + float(value) * n
        self._cnt[key] = self._cnt.get(key, 0) + n

    def update_batch(self, metrics):
        for k in METRIC_NAMES:
            v = metrics[k]
            self.update(k, float(v.double().mean()), int(v.numel()))

    def avg(self, key):
        return self._tot[key] / self._cnt[key]

    def result(self):
        return {k: self.avg(k) for k in self._tot if self._cnt[k]}
