"""Value check of BASELINE.json configs[3]'s per-GPU share EXACTLY as `bench.py --c4` times it: TWO 768x768 images in one
E2E-FT micro-step (training/train.py:470-568), 77-token empty-prompt context, activation recompute ON in the UNet blocks and the frozen
decoder (`--gradient_checkpointing`, training/scripts/train_marigold_e2e_ft_depth.sh:9-11,15; launch training/scripts/multi_gpu.yaml:1-15),
in the reference's precision (fp32, `--mixed_precision no`) and in bf16 compute over fp32 master weights (the other leg `--c4` prints).

Why its own test: the backward at a 96x96 latent takes kernel routes the 576x576 (72x72 latent) tests never reach — the 96-wide `igemm6`
stride-1 dgrad (a width that is a multiple of 32), 256 x 128 `wgrad` tiles at 320 channels over 2 x 9216 pixels, `attn_bwd` / `attn32_bwd`
at 9216 tokens, the 4x4 / stride-2 data gradient of the upsamplers at 768^2 — and recompute re-enters every block from the backward.

The comparison: loss and the ten sampled UNet gradients (first / middle / last layers, the list of tests/test_fullsize_parity_gpu.py)
against torch autograd over the fp32 CPU oracle (`oracle.pipeline_ref.train_forward_ref`, pinned to the reference's step body by
tests/test_reference_wiring_cpu.py).  Bars: fp32 — those of the 576^2 test (loss 1e-3, gradients 5e-3 max-abs / max-ref); bf16 — the bars
the 576^2 test derives from torch's own bf16 noise (profiles/r04_bf16_gradient_noise.md), applied to three rounding draws.
CPU cost of the oracle leg on the GPU box's host: forward + backward of two 768^2 images, about 2 minutes at 64 threads, ~45 GB."""
import copy
import os
import statistics

import pytest
import torch

from oracle import config, pipeline_ref, synth, unet_ref, vae_ref
from util import rel_err

pytestmark = pytest.mark.gpu

from test_fullsize_parity_gpu import GRAD_KEYS, TORCH_BF16_MAX, TORCH_BF16_Q75      # noqa: E402  (the same ten tensors, the same calibration)

RES, IMAGES = 768, 2


def _host_gb():
    try:
        return os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2 ** 30
    except (ValueError, OSError):
        return 0.0


@pytest.fixture(scope="module")
def c4_768(dev):
    """product modules carrying the oracle's seeded network + torch autograd over the fp32 CPU oracle of ONE micro-step of two 768x768 images"""
    from diffusion_e2e_ft_amd import training
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    from diffusion_e2e_ft_amd.vae import AutoencoderKL
    if _host_gb() < 96:
        pytest.skip("the fp32 CPU oracle of two 768x768 images keeps ~45 GB of activations for its backward; this host has %.0f GB" % _host_gb())
    n = torch.get_num_threads()
    torch.set_num_threads(max(n, min(64, os.cpu_count() or 1)))
    try:
        usd = synth.synth_state_dict(unet_ref.unet_param_shapes(config.SD2_UNET), seed=1234)
        vsd = synth.synth_state_dict(vae_ref.vae_param_shapes(config.SD_VAE), seed=4321)
        with torch.device(dev):
            unet = UNet2DConditionModel(in_channels=8)
            vae = AutoencoderKL()
        unet.load_state_dict(usd)
        vae.load_state_dict(vsd)
        text = 0.5 * torch.randn((1, 77, 1024), generator=torch.Generator().manual_seed(9))
        batch = {k: v.cpu() for k, v in training.synthetic_batch(IMAGES, RES, RES, torch.device("cpu"), seed=3).items()}
        sd = dict(usd)
        for k in GRAD_KEYS:
            sd[k] = usd[k].clone().requires_grad_(True)
        loss_ref, _ = pipeline_ref.train_forward_ref(sd, config.SD2_UNET, vsd, config.SD_VAE, batch, text, "depth")
        loss_ref.backward()
    finally:
        torch.set_num_threads(n)
    assert torch.isfinite(loss_ref) and loss_ref.item() > 1e-3
    return unet.eval(), vae.eval(), batch, text, loss_ref.item(), {k: sd[k].grad.detach().clone() for k in GRAD_KEYS}


def _short(k):
    return k.split(".")[0] + ".." + k.split(".")[-2]


def test_config3_c4_768_fp32_micro_step_gradients_with_recompute(dev, c4_768):
    """`bench.py --c4 --dtype fp32`: 2 images @768^2, strict fp32, recompute on — loss and sampled gradients against torch autograd over the oracle"""
    from diffusion_e2e_ft_amd import training
    unet, vae, batch, text, loss_ref, grads_ref = c4_768
    u = copy.deepcopy(unet).train()
    v = copy.deepcopy(vae).eval().requires_grad_(False)
    u.enable_gradient_checkpointing()
    v.enable_gradient_checkpointing()
    torch.cuda.reset_peak_memory_stats()
    loss = training.e2e_ft_loss(u, v, batch, text, "depth")
    loss.backward()
    torch.cuda.synchronize()
    el = abs(loss.item() - loss_ref) / abs(loss_ref)
    named = dict(u.named_parameters())
    errs = {k: rel_err(named[k].grad, grads_ref[k]) for k in GRAD_KEYS}
    print("configs[3] share, 2 x 768^2 fp32 + recompute: loss %.6f (oracle %.6f, rel err %.3e); gradient rel errs %s; peak %.1f GiB"
          % (loss.item(), loss_ref, el, {_short(k): "%.1e" % e for k, e in errs.items()}, torch.cuda.max_memory_allocated() / 2 ** 30))
    assert el <= 1e-3, el
    assert max(errs.values()) <= 5e-3, errs


def test_config3_c4_768_recompute_is_bit_equal_to_stored_activations(dev, c4_768):
    """the same micro-step with and without activation recompute: identical kernels on identical inputs, so loss and gradients are bit-equal
    (at this size the check covers the recomputed 96-wide igemm6 / attention launches that the small-shape recompute test cannot)"""
    from diffusion_e2e_ft_amd import training
    unet, vae, batch, text, _, _ = c4_768
    got = []
    for ckpt in (False, True):
        u = copy.deepcopy(unet).train()
        v = copy.deepcopy(vae).eval().requires_grad_(False)
        if ckpt:
            u.enable_gradient_checkpointing()
            v.enable_gradient_checkpointing()
        loss = training.e2e_ft_loss(u, v, batch, text, "depth")
        loss.backward()
        torch.cuda.synchronize()
        named = dict(u.named_parameters())
        got.append((loss.detach().clone(), {k: named[k].grad.detach().clone() for k in GRAD_KEYS}))
        del u, v, named, loss
    assert torch.equal(got[0][0], got[1][0])
    for k in GRAD_KEYS:
        assert torch.equal(got[0][1][k], got[1][1][k]), k


def test_config3_c4_768_bf16_compute_micro_step_gradients_with_recompute(dev, c4_768):
    """`bench.py --c4` (its default precision: bf16 compute over fp32 master weights, bf16 frozen VAE), recompute on, against the SAME fp32 oracle.
    A bf16 run's gradient error is a random variable (tests/test_fullsize_parity_gpu.py), so three rounding draws — the plain run and two with a 1e-3
    jitter of the latent — against the bars derived from torch's own bf16 distribution at 576^2: median <= 1.25 x its 75th percentile, no draw
    beyond 1.25 x its maximum, cosine >= 0.995, loss within 1e-3 in every draw."""
    from diffusion_e2e_ft_amd import training
    unet, vae, batch, text, loss_ref, grads_ref = c4_768
    orig = training.encode_image

    def draw(seed):
        u = copy.deepcopy(unet).train().set_compute_dtype(torch.bfloat16)
        v = copy.deepcopy(vae).to(torch.bfloat16).eval().requires_grad_(False)
        u.enable_gradient_checkpointing()
        v.enable_gradient_checkpointing()

        def encode(vae_, rgb):
            z = orig(vae_, rgb)
            if seed is None:
                return z
            gj = torch.Generator(device=z.device).manual_seed(seed)
            return (z.float() * (1.0 + 1e-3 * torch.randn(z.shape, generator=gj, device=z.device))).to(z.dtype)

        training.encode_image = encode
        try:
            loss = training.e2e_ft_loss(u, v, batch, text, "depth")
        finally:
            training.encode_image = orig
        loss.backward()
        torch.cuda.synchronize()
        el = abs(loss.item() - loss_ref) / abs(loss_ref)
        named = dict(u.named_parameters())
        l2, cos = {}, {}
        for k in GRAD_KEYS:
            gq, r = named[k].grad.detach().double().cpu().flatten(), grads_ref[k].double().flatten()
            assert named[k].grad.dtype == torch.float32 and torch.isfinite(gq).all(), k
            l2[k] = ((gq - r).norm() / r.norm()).item()
            cos[k] = torch.nn.functional.cosine_similarity(gq, r, dim=0).item()
        print("configs[3] share, 2 x 768^2 bf16 compute + recompute (latent jitter seed %s): loss rel err %.3e; gradient rel L2 errs %s; cosines %s"
              % (seed, el, {_short(k): "%.2e" % e for k, e in l2.items()}, {_short(k): "%.4f" % c for k, c in cos.items()}))
        assert el <= 1e-3, el
        return max(l2.values()), min(cos.values())

    draws = [draw(s) for s in (None, 1, 2)]
    med = statistics.median(d[0] for d in draws)
    print("configs[3] share bf16: worst gradient error per draw %s, median %.3e (bar %.4f); worst cosine per draw %s"
          % (["%.3e" % d[0] for d in draws], med, 1.25 * TORCH_BF16_Q75, ["%.4f" % d[1] for d in draws]))
    assert med <= 1.25 * TORCH_BF16_Q75, draws
    assert max(d[0] for d in draws) <= 1.25 * TORCH_BF16_MAX and min(d[1] for d in draws) >= 0.995, draws
