"""Parity of the HIP path with the CPU oracle ON BASELINE.json's CONFIGURATIONS: the full SD-v2 UNet (866 M
parameters) and SD VAE (84 M), seeded synthetic weights, the same tensors on both sides.

  (i)   configs[0] — one 256x256 image, strict fp32: predicted x0 latent within 1e-3 relative of `oracle.pipeline_ref.single_infer_ref`
        (the north-star tolerance), depth within 2e-3;
  (ii)  configs[1] resolution — one 768x768 image, fp16 compute, against the fp32 oracle at the stated 16-bit tolerance 2e-2;
  (iii) the layer shapes that dominate configs[1] / [2] one by one against torch CPU fp32: conv 128->128 @768^2, fused nearest-upsample
        conv 512->512 @192^2->384^2, split-K conv 1280->1280 @12^2, GroupNorm(+SiLU) 128 ch @768^2, fused self-attention 5 heads x 9216
        tokens, the VAE mid-block attention (one 512-wide head, 9216 tokens) — the full-size code paths no tiny config reaches:
        256x128 tiles on thousands of workgroups with the XCD tile map, 32-bit buffer offsets on GB-sized tensors, split-K at K = 11520,
        9216-key online softmax;
  (iv)  configs[2] resolution — one 576x576 image, E2E-FT micro-step in strict fp32 AND in bf16 compute over fp32 master weights (the
        first training leg of bench.py): loss and sampled UNet gradients (first / middle / last layers) against torch autograd over the
        fp32 oracle (computed once, shared).
The oracle is pinned to the reference's own wiring by tests/test_reference_wiring_cpu.py.  CPU cost of the oracle legs on the GPU
box's host: about 2 s (256^2), 10 s (768^2), 1-2 min (576^2 forward + backward)."""
import os

import pytest
import torch
import torch.nn.functional as TF

from oracle import config, pipeline_ref, unet_ref
from util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cpu_threads():
    n = torch.get_num_threads()
    torch.set_num_threads(max(n, min(64, os.cpu_count() or 1)))     # conftest caps at 8 for the small tests; the oracle legs here are big
    yield
    torch.set_num_threads(n)


@pytest.fixture(scope="module")
def models(dev):
    from diffusion_e2e_ft_amd.synth import init_synthetic_
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    from diffusion_e2e_ft_amd.vae import AutoencoderKL
    with torch.device(dev):
        unet = UNet2DConditionModel(in_channels=8)
        vae = AutoencoderKL()
    init_synthetic_(unet, seed=1234)
    init_synthetic_(vae, seed=4321)
    usd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    vsd = {k: v.detach().float().cpu() for k, v in vae.state_dict().items()}
    assert set(usd) == set(unet_ref.unet_param_shapes(config.SD2_UNET)) and sum(v.numel() for v in usd.values()) == 865_922_244
    g = torch.Generator().manual_seed(0)
    ctx = 0.5 * torch.randn((1, 2, 1024), generator=g)
    return unet.eval(), vae.eval(), usd, vsd, ctx


def _image(res, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (1, 3, res, res), generator=g, dtype=torch.int64).float() / 255.0 * 2.0 - 1.0


def _pipe(unet, vae, dtype, ctx, dev):
    import copy
    from diffusion_e2e_ft_amd.pipeline import MarigoldPipeline
    from diffusion_e2e_ft_amd.scheduler import DDIMScheduler
    if dtype != torch.float32:
        unet, vae = copy.deepcopy(unet).to(dtype), copy.deepcopy(vae).to(dtype)
    pipe = MarigoldPipeline(unet, vae, DDIMScheduler())
    pipe.empty_text_embed = ctx.to(dev, dtype)
    return pipe


def test_config0_256_fp32_latent_and_depth(dev, models, cpu_threads):
    """BASELINE configs[0]: single 256x256 image, fp32 (Marigold/marigold/marigold_pipeline.py:372-478)"""
    from diffusion_e2e_ft_amd import ops
    from diffusion_e2e_ft_amd.modules import to_nchw_view
    from diffusion_e2e_ft_amd.pipeline import _scaled
    unet, vae, usd, vsd, ctx = models
    rgb = _image(256, 5)
    with torch.no_grad():
        want_d, want_x0 = pipeline_ref.single_infer_ref(usd, config.SD2_UNET, vsd, config.SD_VAE, rgb, ctx, return_latent=True)
    pipe = _pipe(unet, vae, torch.float32, ctx, dev)
    with torch.no_grad():
        depth = pipe.single_infer(rgb, 1, False, noise="zeros", normals=False)
        lat = pipe.encode_rgb(rgb.to(dev))
        xin = torch.zeros((1, lat.shape[2], lat.shape[3], 8), device=dev)
        ops.copy_scale(lat.permute(0, 2, 3, 1), xin[..., :4])
        v = pipe.unet(to_nchw_view(xin), pipe.scheduler.timesteps[0], encoder_hidden_states=ctx.to(dev)).sample
        x0 = _scaled(v, -pipe.scheduler.x0_coefficients(999)[1])
    e_x0, e_d = rel_err(x0, want_x0), rel_err(depth, want_d)
    print("256^2 fp32: x0 latent rel err %.3e, depth rel err %.3e" % (e_x0, e_d))
    assert e_x0 <= 1e-3, e_x0
    assert e_d <= 2e-3, e_d


def test_config1_768_fp32_latent_and_depth(dev, models, cpu_threads):
    """The north star's tolerance — 1e-3 relative, fp32, on the latent — AT THE BENCHMARKED RESOLUTION: one 768x768 image through the
    strict-fp32 path (Marigold/marigold/marigold_pipeline.py:372-478): `attn32.hip` at 9216 tokens, the fp32 split-K plans and the fp32 tile map of
    the 768^2 VAE layers end to end, which the 256^2 case (a 32x32 latent) does not reach."""
    unet, vae, usd, vsd, ctx = models
    rgb = _image(768, 7)
    with torch.no_grad():
        want_d, want_x0 = pipeline_ref.single_infer_ref(usd, config.SD2_UNET, vsd, config.SD_VAE, rgb, ctx, return_latent=True)
    pipe = _pipe(unet, vae, torch.float32, ctx, dev)
    with torch.no_grad():
        depth = pipe.single_infer(rgb, 1, False, noise="zeros", normals=False)
        x0 = pipe.predict_latent(rgb)
    torch.cuda.synchronize()
    assert x0.shape == (1, 4, 96, 96) and x0.dtype == torch.float32
    e_x0, e_d = rel_err(x0, want_x0), rel_err(depth, want_d)
    l2 = ((x0.float().cpu() - want_x0).norm() / want_x0.norm()).item()
    print("768^2 fp32: x0 latent max rel err %.3e (rel L2 %.3e), depth rel err %.3e" % (e_x0, l2, e_d))
    assert e_x0 <= 1e-3 and l2 <= 1e-3, (e_x0, l2)
    assert e_d <= 2e-3, e_d


def test_config1_768_fp16_against_fp32_oracle(dev, models, cpu_threads):
    """BASELINE configs[1] resolution, one image: fp16 storage / fp32 accumulation against the fp32 oracle, 2e-2 of the output range"""
    unet, vae, usd, vsd, ctx = models
    rgb = _image(768, 6)
    with torch.no_grad():
        want_d, want_x0 = pipeline_ref.single_infer_ref(usd, config.SD2_UNET, vsd, config.SD_VAE, rgb, ctx, return_latent=True)
        want_n = pipeline_ref.decode_ref(vsd, config.SD_VAE, want_x0)
        want_n = want_n / (torch.norm(want_n, p=2, dim=1, keepdim=True) + 1e-5)
    pipe = _pipe(unet, vae, torch.float16, ctx, dev)
    with torch.no_grad():
        depth = pipe.single_infer(rgb, 1, False, noise="zeros", normals=False)
        normal = pipe.single_infer(rgb, 1, False, noise="zeros", normals=True)
    e = rel_err(depth.float(), want_d)
    mae = (depth.float().cpu() - want_d).abs().mean().item()
    cos = (TF.normalize(normal.float().cpu(), dim=1) * want_n).sum(1).clamp(-1, 1)
    ang = torch.rad2deg(torch.acos(cos)).mean().item()
    print("768^2 fp16: depth max rel err %.3e, mean abs err %.3e; normals mean angle %.3f deg" % (e, mae, ang))
    assert e <= 2e-2 and mae <= 2e-3, (e, mae)
    assert ang <= 1.0, ang


# ---- (iii) the dominant layer shapes, one by one -------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_batched_time_and_context_projections_are_bit_identical(dev, models, dtype):
    """Inference computes the 22 `time_emb_proj` rows and the 16 cross-attention key / value projections as two GEMMs over concatenated weights
    (unet.py::_batch_small_gemms).  Same dot products, element for element: the full SD-v2 UNet output must equal the per-layer launches
    bit for bit (fp32: the time projections only — the fp32 attention path projects inside `attention_unfused`)."""
    import copy
    unet, _, _, _, ctx = models
    m = copy.deepcopy(unet).to(dtype).eval()
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 8, 32, 32), generator=g).to(dev, dtype)
    c = ctx.to(dev, dtype).expand(2, -1, -1).contiguous()
    with torch.no_grad():
        a = m(x, 999, c).sample
        fam = m.__dict__["_small_gemm_family"]
        assert len(fam[0]) == 22 and len(fam[1]) == 16
        assert all(not any(k.startswith(("_rowadd", "_kv")) for k in r.__dict__) for r in fam[0] + fam[1])   # the slices travel as call arguments: nothing parked on a module
        m._batch_small_gemms = lambda temb_act, ctx_, *unused: (temb_act, ctx_)
        b = m(x, 999, c).sample
    assert torch.isfinite(a.float()).all() and torch.equal(a, b)


def _conv_big(dev, dtype, B, Ci, Co, H, W, stride=1, up_to=None, seed=0, tol=None):
    from diffusion_e2e_ft_amd import ops
    from util import TOL, nhwc, pack_conv_weight, q, to_nchw
    g = torch.Generator().manual_seed(seed)
    x = q(torch.randn(B, Ci, H, W, generator=g), dtype)
    w = q(torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5, dtype)
    b = q(torch.randn(Co, generator=g), dtype)
    xin = x if up_to is None else TF.interpolate(x, size=up_to, mode="nearest")
    ref = TF.conv2d(xin, w, b, stride=stride, padding=1)
    out = ops.conv2d(nhwc(x, dtype, dev), pack_conv_weight(w, dtype, dev), b.to(dtype).to(dev), Co, 3, 3, stride, (1, 1, 1, 1), up_to=up_to)
    e = rel_err(to_nchw(out), ref)
    assert e <= (tol or TOL[dtype]), e
    return e


def test_conv_128_at_768(dev, cpu_threads):
    """VAE 128-channel level at full resolution: 2304 tiles of 256x128, K = 1152 (14 % of the inference step)"""
    print("conv 128->128 @768^2 fp16 rel err %.2e" % _conv_big(dev, torch.float16, 1, 128, 128, 768, 768))


def test_conv_upsample_512_at_192_to_384(dev, cpu_threads):
    """decoder up-block: nearest 2x fused into the gather, K = 4608"""
    print("upsample conv 512->512 @192->384 fp16 rel err %.2e" % _conv_big(dev, torch.float16, 1, 512, 512, 192, 192, up_to=(384, 384), seed=1))


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_conv_splitk_1280_at_12(dev, dtype, cpu_threads):
    """UNet mid level at 768^2 input with batch 8: 1152 pixels, K = 11520, split-K over filter-tap rows"""
    print("split-K conv 1280->1280 @12^2 %s rel err %.2e" % (dtype, _conv_big(dev, dtype, 8, 1280, 1280, 12, 12, seed=2)))


def test_groupnorm_128_at_768(dev, cpu_threads):
    from diffusion_e2e_ft_amd import ops
    from util import nhwc, q, to_nchw
    dtype = torch.float16
    g = torch.Generator().manual_seed(3)
    x = q(torch.randn(2, 128, 768, 768, generator=g) * 1.5 + torch.randn(1, 128, 1, 1, generator=g), dtype)
    ga, be = q(1 + 0.3 * torch.randn(128, generator=g), dtype), q(0.3 * torch.randn(128, generator=g), dtype)
    ref = TF.silu(TF.group_norm(x, 32, ga, be, 1e-6))
    out = ops.groupnorm(nhwc(x, dtype, dev), ga.to(dtype).to(dev), be.to(dtype).to(dev), 32, 1e-6, True)
    e = rel_err(to_nchw(out), ref)
    print("GroupNorm+SiLU 128 ch @768^2 fp16 rel err %.2e" % e)
    assert e <= 4.5e-3, e


def test_attention_5_heads_9216_tokens(dev, cpu_threads):
    """UNet level 0 self-attention at 768^2: 9216 queries x 9216 keys, 5 heads of 64"""
    from diffusion_e2e_ft_amd import ops
    from util import q
    dtype = torch.float16
    g = torch.Generator().manual_seed(4)
    qq, kk, vv = (q(torch.randn(1, 9216, 320, generator=g), dtype) for _ in range(3))
    kk[0, 9000] = 3.0 * qq[0, 17]              # one dominant key late in the sequence: exercises the running-max rescale at full length
    sp = lambda t: t.reshape(1, 9216, 5, 64).transpose(1, 2)
    ref = TF.scaled_dot_product_attention(sp(qq), sp(kk), sp(vv)).transpose(1, 2).reshape(1, 9216, 320)
    out = ops.attention(qq.to(dtype).to(dev), kk.to(dtype).to(dev), vv.to(dtype).to(dev), 5, 64 ** -0.5)
    e = rel_err(out, ref)
    print("attention 5 x 9216 x 9216 fp16 rel err %.2e" % e)
    assert e <= 4.5e-3, e


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_vae_midblock_attention_512_wide_9216_tokens(dev, dtype, cpu_threads):
    """AutoencoderKL mid-block attention at 768^2 (unet_2d_blocks.py:589-601): one head of width 512, biases, 9216 tokens"""
    from diffusion_e2e_ft_amd.modules import Attention
    from util import q
    g = torch.Generator().manual_seed(5)
    C, N = 512, 9216
    m = Attention(C, 1, bias=True)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(q(torch.randn(p.shape, generator=g) * (C ** -0.5 if p.dim() == 2 else 0.1), dtype))
    x = q(torch.randn(1, N, C, generator=g), dtype)
    res = q(torch.randn(1, N, C, generator=g), dtype)
    with torch.no_grad():
        qv, kv, vv = (TF.linear(x, getattr(m, n).weight, getattr(m, n).bias) for n in ("to_q", "to_k", "to_v"))
        a = TF.scaled_dot_product_attention(qv[:, None], kv[:, None], vv[:, None])[:, 0]
        ref = TF.linear(a, m.to_out[0].weight, m.to_out[0].bias) + res
        out = m.to(dev, dtype)(x.to(dev, dtype), residual=res.to(dev, dtype))
    e = rel_err(out.float(), ref)
    print("VAE attention d=512 N=9216 %s rel err %.2e" % (dtype, e))
    assert e <= (6e-3 if dtype == torch.float16 else 4e-2), e


# ---- (iv) training micro-step at the configs[2] resolution ------------------------------------------------------------------------
# (the list scripts/bf16_localise.py samples: the torch-bf16 calibration below is the worst error over exactly these ten tensors)
GRAD_KEYS = ["conv_in.weight", "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight", "down_blocks.1.resnets.0.conv1.weight",
             "mid_block.attentions.0.proj_in.weight", "mid_block.resnets.1.conv2.weight", "up_blocks.1.resnets.0.conv_shortcut.weight",
             "up_blocks.2.attentions.1.transformer_blocks.0.attn2.to_k.weight", "up_blocks.3.resnets.2.norm2.weight", "conv_norm_out.bias", "conv_out.weight"]


# ---- (iv) configs[2] resolution: the E2E-FT micro-step ------------------------------------------------------------------------------------------
# The instance is the one the torch-bf16 calibration was taken on (profiles/r04_bf16_*): the ORACLE's seeded network (oracle.synth.synth_state_dict, per-key CPU
# generators) loaded into the product modules, the sample from the CPU generator.  Round 4 found that the gradient error of a bf16 run is a property of the
# instance first and of the implementation second (profiles/r04_bf16_gradient_noise.md): the reference's loss is an L1 over a least-squares-aligned prediction
# (training/util/loss.py:13-47), every prediction error that moves a residual through zero flips that pixel's gradient, and the smooth part of the gradient has
# coefficients that are differences of nearly cancelling sums of signs — so two random networks differ by 5x in how much gradient error the same 3 % forward
# error produces (0.05 on this instance, 0.04 ... 0.17 and 0.14 ... 0.31 on the two device-seeded instances of round 3), for torch's own bf16 as for the HIP path.
@pytest.fixture(scope="module")
def calibrated_576(dev, cpu_threads):
    """product modules carrying the oracle's network + torch autograd over the fp32 CPU oracle of ONE 576x576 micro-step (training/train.py:470-556): loss and the
    ten sampled gradients.  Computed once (1-2 minutes of host time) and shared by the fp32 and the bf16-compute tests below."""
    from oracle import synth, vae_ref
    from diffusion_e2e_ft_amd import training
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    from diffusion_e2e_ft_amd.vae import AutoencoderKL
    usd = synth.synth_state_dict(unet_ref.unet_param_shapes(config.SD2_UNET), seed=1234)
    vsd = synth.synth_state_dict(vae_ref.vae_param_shapes(config.SD_VAE), seed=4321)
    with torch.device(dev):
        unet = UNet2DConditionModel(in_channels=8)
        vae = AutoencoderKL()
    unet.load_state_dict(usd)
    vae.load_state_dict(vsd)
    g = torch.Generator().manual_seed(9)
    text = 0.5 * torch.randn((1, 77, 1024), generator=g)
    batch = {k: v.cpu() for k, v in training.synthetic_batch(1, 576, 576, torch.device("cpu"), seed=3).items()}
    sd = dict(usd)
    for k in GRAD_KEYS:
        sd[k] = usd[k].clone().requires_grad_(True)
    loss_ref, _ = pipeline_ref.train_forward_ref(sd, config.SD2_UNET, vsd, config.SD_VAE, batch, text, "depth")
    loss_ref.backward()
    assert abs(loss_ref.item() - 0.396215) < 2e-5, loss_ref.item()        # the instance of the calibration (its fp32 loss)
    return unet.eval(), vae.eval(), batch, text, loss_ref.item(), {k: sd[k].grad.detach().clone() for k in GRAD_KEYS}


def test_config2_576_fp32_micro_step_gradients(dev, calibrated_576):
    """BASELINE configs[2] resolution (576x576, 77-token context, `--mixed_precision no`): loss and sampled UNet gradients of one
    micro-step (training/train.py:470-556) against torch autograd over the oracle"""
    import copy
    from diffusion_e2e_ft_amd import training
    unet, vae, batch, text, loss_ref, grads_ref = calibrated_576
    u = copy.deepcopy(unet).train()
    v = vae.requires_grad_(False)
    loss = training.e2e_ft_loss(u, v, batch, text, "depth")
    loss.backward()
    torch.cuda.synchronize()
    el = abs(loss.item() - loss_ref) / abs(loss_ref)
    named = dict(u.named_parameters())
    errs = {k: rel_err(named[k].grad, grads_ref[k]) for k in GRAD_KEYS}
    print("576^2 fp32 micro-step: loss rel err %.3e; gradient rel errs %s" % (el, {k.split(".")[0] + ".." + k.split(".")[-2]: "%.1e" % e for k, e in errs.items()}))
    assert el <= 1e-3, el
    assert max(errs.values()) <= 5e-3, errs


# torch bf16 on this instance (the CPU oracle with a bf16 state dict and bf16 activations against its own fp32 run, 13 draws: profiles/r04b_bf16_localise_cpu.tsv):
# worst sampled parameter-gradient error per draw min 0.044, quartiles 0.046 / 0.049 / 0.054, max 0.065
TORCH_BF16_Q75, TORCH_BF16_MAX = 0.054, 0.065


def test_config2_576_bf16_compute_micro_step_gradients(dev, calibrated_576):
    """The training leg bench.py reports first (`train_step`: bf16 compute over fp32 master weights, bf16 frozen VAE) at the configs[2] resolution against the
    SAME fp32 oracle, on the instance the torch-bf16 calibration was taken on.  The error of a bf16 run is a random variable (a perturbation of the latent far
    below one bf16 ulp re-draws the roundings), so the bar is on the distribution, and it is DERIVED FROM TORCH'S: five draws (the plain run
    and four with a 1e-3 jitter of the latent) — loss within 1e-3 in every draw, the MEDIAN of the per-draw worst gradient error <= 1.25 x torch-bf16's 75th
    percentile, no draw beyond 1.25 x torch-bf16's maximum.  Measured on the round-4 build, 13 draws (profiles/r04b_bf16_localise_hip.tsv): min 0.051, quartiles
    0.057 / 0.060 / 0.066, max 0.074 — 1.2 x torch at the median with the same spread; the forward activations are CLOSER to fp32 than torch's at every block
    boundary (0.93-0.95 x), the excess enters at the loss gradient (profiles/r04_bf16_gradient_noise.md)."""
    import copy
    import statistics
    from diffusion_e2e_ft_amd import training
    unet, vae, batch, text, loss_ref, grads_ref = calibrated_576
    short = lambda k: k.split(".")[0] + ".." + k.split(".")[-2]
    orig = training.encode_image

    def draw(seed):
        u = copy.deepcopy(unet).train().set_compute_dtype(torch.bfloat16)
        v = copy.deepcopy(vae).to(torch.bfloat16).eval().requires_grad_(False)

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
        print("576^2 bf16-compute micro-step (latent jitter seed %s): loss rel err %.3e; gradient rel L2 errs %s; cosines %s"
              % (seed, el, {short(k): "%.2e" % e for k, e in l2.items()}, {short(k): "%.4f" % c for k, c in cos.items()}))
        assert el <= 1e-3, el
        return max(l2.values()), min(cos.values())

    draws = [draw(s) for s in (None, 1, 2, 3, 4)]
    med = statistics.median(d[0] for d in draws)
    print("576^2 bf16-compute micro-step: worst gradient error per draw %s, median %.3e (bar %.4f = 1.25 x torch-bf16's 75th percentile); worst cosine per draw %s"
          % (["%.3e" % d[0] for d in draws], med, 1.25 * TORCH_BF16_Q75, ["%.4f" % d[1] for d in draws]))
    assert med <= 1.25 * TORCH_BF16_Q75, draws
    assert max(d[0] for d in draws) <= 1.25 * TORCH_BF16_MAX and min(d[1] for d in draws) >= 0.995, draws
