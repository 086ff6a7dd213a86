"""Depth evaluation on the device (csrc/evalmetrics.hip, diffusion_e2e_ft_amd/evaluate.py) against the REFERENCE'S functions: the fixture
tests/golden/eval_golden.pt holds what Marigold/src/util/alignment.py + metric.py return for the chain of Marigold/eval.py:172-209 on seeded
synthetic triples (tests/golden/make_eval_golden.py): least-squares alignment (full resolution, nearest-down-sampled to max_resolution,
disparity space), clipping, the ten metrics.  Tolerances: scale / shift 2e-4 relative (numpy solves the float32 system through an SVD,
the device the fp64 normal equations), metrics 1e-4 relative / 1e-5 absolute; BASELINE.json's acceptance bar is AbsRel within 1e-3."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_eval_golden import CASES, NAMES, SETTINGS, eval_case  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = torch.load(os.path.join(HERE, "golden", "eval_golden.pt"), weights_only=False)


@pytest.mark.parametrize("si", range(len(SETTINGS)))
def test_depth_metrics_match_reference_functions(dev, si):
    from diffusion_e2e_ft_amd import evaluate
    s = SETTINGS[si]
    for ci, c in enumerate(CASES):
        pred, gt, mask = eval_case(**c)
        out = evaluate.depth_metrics(pred.to(dev), gt.to(dev), mask.to(dev), alignment=s["alignment"], min_depth=1e-3, max_depth=80.0,
                                     alignment_max_res=s["max_res"], return_aligned=True)
        want = GOLD[(ci, si)]
        assert abs(out["scale"].item() - want["scale"]) <= 2e-4 * abs(want["scale"]) + 1e-6, (ci, out["scale"].item(), want["scale"])
        assert abs(out["shift"].item() - want["shift"]) <= 2e-4 * abs(want["shift"]) + 2e-4
        for i, n in enumerate(NAMES):
            got, w = out[n].item(), want["metrics"][i].item()
            assert abs(got - w) <= 1e-4 * abs(w) + 1e-5, (ci, s, n, got, w)
        a = out["aligned"][0].cpu()[::7, ::5]
        assert ((a - want["aligned_sample"]).abs() / want["aligned_sample"].abs().clamp_min(1e-3)).max().item() < 5e-4
    assert list(evaluate.METRIC_NAMES) == list(NAMES)


def test_batched_evaluation_tracker_and_alignment_surface(dev):
    """a whole batch in one launch sequence = the per-image calls; MetricTracker averages per image like metric.py:9-31;
    align_depth_least_square keeps the reference's return convention; the result is bit-reproducible"""
    from diffusion_e2e_ft_amd import evaluate
    trip = [eval_case(seed=10 + i, H=48, W=64) for i in range(4)]
    pred, gt, mask = (torch.stack([t[k] for t in trip]).to(dev) for k in range(3))
    allb = evaluate.depth_metrics(pred, gt, mask)
    again = evaluate.depth_metrics(pred, gt, mask)
    tr = evaluate.MetricTracker(*evaluate.METRIC_NAMES)
    for i in range(4):
        one = evaluate.depth_metrics(pred[i], gt[i], mask[i])
        for n in evaluate.METRIC_NAMES:
            assert one[n].item() == allb[n][i].item() == again[n][i].item()
            tr.update(n, one[n].item())
    tr2 = evaluate.MetricTracker(*evaluate.METRIC_NAMES)
    tr2.update_batch(allb)
    assert abs(tr.avg("abs_relative_difference") - tr2.avg("abs_relative_difference")) < 1e-7
    aligned, scale, shift = evaluate.align_depth_least_square(gt[0], pred[0], mask[0], return_scale_shift=True)
    assert aligned.shape == pred[0].shape and abs(scale.item() - allb["scale"][0].item()) < 1e-6
    assert torch.allclose(aligned, pred[0] * scale + shift)
    # end to end: the acceptance number of BASELINE.json for a prediction that IS an affine map of the ground truth is 0
    perfect = evaluate.depth_metrics((gt[0] - 1.0) / 25.0, gt[0], mask[0])
    assert perfect["abs_relative_difference"].item() < 1e-5 and perfect["delta1_acc"].item() == 1.0


def test_absrel_acceptance_on_synthetic_hypersim(dev):
    """BASELINE.json: "depth AbsRel within 1e-3 of the reference on Hypersim-val synthetic" — device evaluation vs the host oracle chain
    (oracle/metric_ref.py, pinned to the reference in tests/test_oracle_pins.py) on the synthetic batch of SURVEY.md §8d"""
    from diffusion_e2e_ft_amd import evaluate, training
    from oracle import metric_ref
    b = training.synthetic_batch(2, 96, 128, dev, seed=4)
    gt = (b["metric"][:, 0] + 1.5) * 6.0                      # metres, positive
    g = torch.Generator(device=dev).manual_seed(1)
    pred = ((gt - gt.amin()) / (gt.amax() - gt.amin()) + 0.03 * torch.randn(gt.shape, generator=g, device=dev)).clamp(0, 1)
    mask = b["val_mask"][:, 0]
    got = evaluate.depth_metrics(pred, gt, mask, min_depth=1e-3, max_depth=80.0)["abs_relative_difference"]
    for i in range(2):
        want = metric_ref.aligned_absrel_ref(pred[i].cpu(), gt[i].cpu(), mask[i].cpu())
        assert abs(got[i].item() - want.item()) < 1e-4, (got[i].item(), want.item())
