"""TEST INFRASTRUCTURE (see oracle/__init__.py).  CPU restatement of the evaluation arithmetic behind BASELINE.json's "depth AbsRel
within 1e-3 of the reference": least-squares scale/shift alignment of an affine-invariant prediction to metric depth
(/root/reference/Marigold/src/util/alignment.py:9-58) and the AbsRel metric (/root/reference/Marigold/src/util/metric.py:35-46).
Pinned against the reference's own functions in tests/test_oracle_pins.py when /root/reference is present."""
import numpy as np
import torch


def align_depth_least_square_ref(gt, pred, valid_mask):
    """gt, pred, valid_mask: [H,W] numpy arrays -> (aligned_pred, scale, shift); alignment.py:37-50"""
    g = gt[valid_mask].reshape(-1, 1).astype(np.float64)
    p = pred[valid_mask].reshape(-1, 1).astype(np.float64)
    A = np.concatenate([p, np.ones_like(p)], axis=-1)
    X = np.linalg.lstsq(A, g, rcond=None)[0]
    scale, shift = float(X[0, 0]), float(X[1, 0])
    return pred * scale + shift, scale, shift


def abs_relative_difference_ref(output, target, valid_mask=None):
    """torch tensors [..., H, W]; metric.py:35-46"""
    d = torch.abs(output - target) / target
    if valid_mask is not None:
        d = torch.where(valid_mask, d, torch.zeros_like(d))
        n = valid_mask.sum((-1, -2))
    else:
        n = output.shape[-1] * output.shape[-2]
    return (torch.sum(d, (-1, -2)) / n).mean()


def aligned_absrel_ref(pred, gt, valid_mask, clip=(1e-3, 80.0)):
    """the evaluation chain of Marigold/eval.py for one image: LS alignment, clip to the dataset range, AbsRel"""
    aligned, _, _ = align_depth_least_square_ref(gt.numpy(), pred.numpy(), valid_mask.numpy())
    aligned = torch.from_numpy(np.clip(aligned, clip[0], clip[1])).to(gt.dtype)
    return abs_relative_difference_ref(aligned, gt, valid_mask)
