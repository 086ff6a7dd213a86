"""CPU tests for test-time ensembling: the oracle restatement (oracle/ensemble_ref.py) is PINNED against outputs of the reference's
own functions (tests/golden/ensemble_golden.pt, made by tests/golden/make_ensemble_golden.py) and against the live reference
when /root/reference is present; plus the host-side algebra of diffusion_e2e_ft_amd.ensemble (Gram form of the pairwise term)."""
import importlib.util
import os
import warnings

import numpy as np
import pytest
import torch

import golden_cases as gc
from oracle import ensemble_ref

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = torch.load(os.path.join(HERE, "golden", "ensemble_golden.pt"))
REF_DEPTH = "/root/reference/Marigold/marigold/util/ensemble.py"


@pytest.mark.parametrize("name", list(gc.ENSEMBLE_DEPTH_CASES))
def test_depth_oracle_matches_reference_golden(name):
    case = gc.ENSEMBLE_DEPTH_CASES[name]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")   # scipy: the forward-difference step underflows in float32 (see ensemble.py docstring)
        aligned, unc = ensemble_ref.ensemble_depths_ref(gc.ensemble_depth_stack(**case["stack"]), **case["kw"])
    g = GOLD["depth"][name]
    assert torch.equal(aligned, g["aligned"]) and torch.equal(unc, g["uncertainty"])   # same torch ops in the same order: bit-exact


@pytest.mark.parametrize("name,kw", [("n6", {}), ("n3", dict(n=3, H=17, W=29, seed=39))])
def test_normals_oracle_matches_reference_golden(name, kw):
    pred, none = ensemble_ref.ensemble_normals_ref(gc.ensemble_normal_stack(**kw))
    assert none is None
    assert torch.equal(pred, GOLD["normals"][name])


@pytest.mark.skipif(not os.path.exists(REF_DEPTH), reason="reference tree not present")
def test_depth_oracle_against_live_reference():
    spec = importlib.util.spec_from_file_location("ref_ensemble_live", REF_DEPTH)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    stack = gc.ensemble_depth_stack(n=6, H=21, W=30, seed=77)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for kw in ({}, dict(reduction="mean"), dict(max_res=16)):
            a, u = m.ensemble_depths(stack.clone(), **kw)
            b, v = ensemble_ref.ensemble_depths_ref(stack.clone(), **kw)
            assert torch.equal(a, b) and torch.equal(u, v), kw


def test_reference_optimiser_is_inert_in_float32():
    """Documents the behaviour the product mirrors: BFGS over a float32 x0 gets a NaN numerical gradient and returns x0, so the
    result is the closed-form min-max alignment."""
    stack = gc.ensemble_depth_stack(n=5, seed=41)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a, _ = ensemble_ref.ensemble_depths_ref(stack, max_iter=50)
    flat = stack.reshape(5, -1)
    s = 1.0 / (flat.max(1).values - flat.min(1).values)
    al = stack * s.view(-1, 1, 1) + (-s * flat.min(1).values).view(-1, 1, 1)
    med = al.median(0).values
    assert torch.allclose(a, (med - med.min()) / (med.max() - med.min()), atol=1e-6)


def test_gram_form_of_the_pairwise_term():
    """host algebra of diffusion_e2e_ft_amd.ensemble._pair_term == direct sum over pairs and pixels"""
    import importlib
    ens = importlib.import_module("diffusion_e2e_ft_amd.ensemble")
    g = torch.Generator().manual_seed(3)
    d = torch.randn(6, 500, generator=g, dtype=torch.float64)
    s = torch.rand(6, generator=g, dtype=torch.float64) + 0.5
    t = torch.randn(6, generator=g, dtype=torch.float64)
    a = d * s[:, None] + t[:, None]
    i, j = torch.triu_indices(6, 6, offset=1)
    direct = float(((a[i] - a[j]) ** 2).sum())
    got = ens._pair_term((d @ d.T).numpy(), d.sum(1).numpy(), 500, s.numpy(), t.numpy())
    assert abs(got - direct) <= 1e-9 * abs(direct)
