"""CPU tests pinning the oracle (oracle/) — the checker of the GPU parity tests.

 * hooks (losses, LR schedule, conv_in replacement): PINNED against the reference's own Python modules imported from
   /root/reference/training/util when that tree is present, and against tests/golden/hooks_golden.pt generated from them.
 * UNet / VAE / pipeline restatements: parity UNPINNED at the diffusers boundary (no golden vectors exist in the reference,
   diffusers is not installable here).  Structural pins: parameter counts and key set of the published models, scheduler
   constants, analytic known-answer tests, and a regression pin of the oracle's own outputs (model_golden.pt)."""
import importlib.util
import math
import os

import pytest
import torch
import torch.nn.functional as F

import golden_cases as gc
from oracle import config, unet_ref, vae_ref, pipeline_ref, losses_ref, synth

HERE = os.path.dirname(os.path.abspath(__file__))
REF_UTIL = "/root/reference/training/util"
HOOKS = torch.load(os.path.join(HERE, "golden", "hooks_golden.pt"))


def _numel(shapes):
    return sum(math.prod(s) for s in shapes.values())


def test_param_counts_match_published_models():
    assert _numel(unet_ref.unet_param_shapes(dict(config.SD2_UNET, in_channels=4))) == 865_910_724  # SD-v2 UNet
    assert _numel(unet_ref.unet_param_shapes(config.SD2_UNET)) == 865_922_244                        # + 8-channel conv_in
    assert _numel(unet_ref.unet_param_shapes(config.GEOWIZARD_UNET)) == 861_186_244
    assert _numel(vae_ref.vae_param_shapes(config.SD_VAE)) == 83_653_863                              # SD VAE
    assert len(unet_ref.unet_param_shapes(config.SD2_UNET)) == 686
    assert len(unet_ref.unet_param_shapes(config.GEOWIZARD_UNET)) == 690


def test_scheduler_constants():
    ac = pipeline_ref.alphas_cumprod()
    assert abs(ac[999].item() - 0.00466010) < 1e-7 and abs(ac[0].item() - 0.99914998) < 1e-7
    assert abs(ac[999].sqrt().item() - 0.06826489) < 1e-7 and abs((1 - ac[999]).sqrt().item() - 0.99766725) < 1e-7
    assert list(pipeline_ref.trailing_timesteps(1)) == [999] and list(pipeline_ref.trailing_timesteps(2)) == [999, 499]
    v = torch.randn(2, 4, 3, 3)
    assert torch.allclose(pipeline_ref.v_to_x0(v, torch.zeros_like(v), 999), -0.99766725 * v, atol=1e-6)


def test_lr_schedule_golden_and_kat():
    vals = [losses_ref.iter_exponential_ref(int(i), 20000, 0.01, 100) for i in HOOKS["lr_iters"]]
    assert torch.allclose(torch.tensor(vals, dtype=torch.float64), HOOKS["lr_values"], rtol=1e-12)
    kat = {0: 0.0, 50: 0.5, 100: 1.0, 10050: 0.1, 20000: 0.01, 30000: 0.01}  # SURVEY.md §8c observed values
    for i, v in kat.items():
        assert abs(losses_ref.iter_exponential_ref(i, 20000, 0.01, 100) - v) < 1e-9


def test_losses_match_reference_golden():
    pred, tgt, mask = gc.ssi_inputs()
    assert torch.allclose(losses_ref.ssi_loss_ref(pred, tgt, mask), HOOKS["ssi_loss"], rtol=1e-6)
    s, t = losses_ref.compute_scale_and_shift_masked_ref(pred.squeeze(1), tgt.squeeze(1), mask.squeeze(1))
    assert torch.allclose(s, HOOKS["ssi_scale"]) and torch.allclose(t, HOOKS["ssi_shift"])
    n, nt, m = gc.angular_inputs()
    assert torch.allclose(losses_ref.angular_loss_ref(n, nt, m), HOOKS["angular_loss"], rtol=1e-6)


def test_loss_kats():
    pred, tgt, mask = gc.ssi_inputs()
    assert losses_ref.ssi_loss_ref(2.0 * tgt - 0.3, tgt, mask).item() < 1e-6      # affine relation => 0
    n, _, m = gc.angular_inputs()
    assert losses_ref.angular_loss_ref(n, n, m).item() < 1e-3                      # identical unit vectors => ~0


def test_conv_in_replacement_matches_reference_golden():
    w, b = losses_ref.replace_conv_in_ref(HOOKS["conv_in_w0"], HOOKS["conv_in_b0"], repeat=2)
    assert torch.equal(w, HOOKS["conv_in_w"]) and torch.equal(b, HOOKS["conv_in_b"]) and int(HOOKS["conv_in_cfg"]) == 8


@pytest.mark.skipif(not os.path.isdir(REF_UTIL), reason="reference tree not present")
def test_hooks_against_live_reference():
    def imp(name):
        spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF_UTIL, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    loss, lr = imp("loss"), imp("lr_scheduler")
    g = torch.Generator().manual_seed(5)
    for _ in range(3):
        tgt = torch.rand(2, 1, 24, 32, generator=g)
        pred = torch.randn(2, 1, 24, 32, generator=g)
        mask = torch.rand(2, 1, 24, 32, generator=g) > 0.3
        assert torch.allclose(losses_ref.ssi_loss_ref(pred, tgt, mask), loss.ScaleAndShiftInvariantLoss()(pred, tgt, mask), rtol=1e-6)
        n = F.normalize(torch.randn(2, 3, 24, 32, generator=g), dim=1)
        t = F.normalize(torch.randn(2, 3, 24, 32, generator=g), dim=1)
        assert torch.allclose(losses_ref.angular_loss_ref(n, t, mask), loss.AngularLoss()(n, t, mask), rtol=1e-6)
    s = lr.IterExponential(1000, 0.05, 10)
    for i in (0, 5, 10, 500, 999, 1000, 2000):
        assert abs(s(i) - losses_ref.iter_exponential_ref(i, 1000, 0.05, 10)) < 1e-12


def test_oracle_leaf_kats():
    """analytic known-answer tests of the restated leaf semantics"""
    # sinusoid: [cos | sin] order with flip_sin_to_cos, f_0 = 1
    e = unet_ref.timestep_sinusoid(torch.tensor([999]), 320)
    assert abs(e[0, 0].item() - math.cos(999.0)) < 1e-4 and abs(e[0, 160].item() - math.sin(999.0)) < 1e-4
    # attention with a single key: softmax == 1 => output = to_out(v) for every query
    C = 64
    sd = {"a.to_q.weight": torch.randn(C, C), "a.to_k.weight": torch.randn(C, 32), "a.to_v.weight": torch.randn(C, 32),
          "a.to_out.0.weight": torch.eye(C), "a.to_out.0.bias": torch.zeros(C)}
    x, ctx = torch.randn(2, 10, C), torch.randn(2, 1, 32)
    out = unet_ref.attention(sd, "a", x, ctx, heads=1)
    assert torch.allclose(out, (ctx @ sd["a.to_v.weight"].t()).expand(2, 10, C), atol=1e-5)
    # joint attention: both task halves see the same keys => permuting the halves of K/V rows leaves outputs' halves swapped
    sdj = {k: v for k, v in sd.items()}
    sdj["a.to_k.weight"], sdj["a.to_v.weight"] = torch.randn(C, C), torch.randn(C, C)
    xj = torch.randn(4, 6, C)
    o1 = unet_ref.attention(sdj, "a", xj, None, heads=1, joint=True)
    o2 = unet_ref.attention(sdj, "a", torch.cat([xj[2:], xj[:2]]), None, heads=1, joint=True)
    assert torch.allclose(o1, torch.cat([o2[2:], o2[:2]]), atol=1e-5)
    # VAE downsample is the asymmetric (0,1,0,1) pad: output size floor((H+1-3)/2)+1
    vsd = gc.tiny_vae_sd()
    m = vae_ref.encoder_forward(vsd, config.TINY_VAE, torch.zeros(1, 3, 40, 56))
    assert m.shape == (1, 8, 5, 7)


def test_oracle_reproduces_model_golden():
    """regression pin of the oracle itself (and a determinism check of the seeded synthetic weights)"""
    gold = torch.load(os.path.join(HERE, "golden", "model_golden.pt"))
    with torch.no_grad():
        for name in ("unet_16x16", "vae", "geo_unet"):
            now = gc.MODEL_CASES[name]()
            for k, v in now.items():
                assert torch.allclose(v, gold[name][k], rtol=1e-4, atol=1e-5), (name, k)


def test_absrel_metric_restatement_matches_reference():
    """oracle/metric_ref.py vs the reference's alignment.py / metric.py on seeded data"""
    import importlib.util
    import numpy as np
    from oracle import metric_ref
    base = "/root/reference/Marigold/src/util"
    if not os.path.isdir(base):
        pytest.skip("reference tree not present")

    def imp(name):
        spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(base, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    al, me = imp("alignment"), imp("metric")
    g = torch.Generator().manual_seed(5)
    gt = torch.rand(48, 64, generator=g) * 9 + 1
    pred = (gt - 2.0) / 7.0 + 0.03 * torch.randn(48, 64, generator=g)
    mask = torch.rand(48, 64, generator=g) > 0.1
    a_ref, s_ref, t_ref = al.align_depth_least_square(gt.numpy(), pred.numpy(), mask.numpy())
    a, s_, t_ = metric_ref.align_depth_least_square_ref(gt.numpy(), pred.numpy(), mask.numpy())
    assert abs(s_ - float(s_ref)) < 1e-5 * abs(float(s_ref)) and abs(t_ - float(t_ref)) < 1e-5 and np.allclose(a, a_ref, rtol=1e-5, atol=1e-5)
    x = torch.from_numpy(a_ref).float()
    assert torch.allclose(metric_ref.abs_relative_difference_ref(x, gt, mask), me.abs_relative_difference(x.clone(), gt, mask), rtol=1e-6)
    assert torch.allclose(metric_ref.abs_relative_difference_ref(x, gt), me.abs_relative_difference(x.clone(), gt), rtol=1e-6)
