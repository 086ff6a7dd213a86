"""Value checks of the HIP path on the EXACT configurations the JSON lines of bench.py are printed for:

  (i)   BASELINE.json configs[1] as benchmarked — EIGHT distinct 768x768 images, fp16, ONE `single_infer` call (batch 8: the 256-channel
        768^2 activations are 2.42 GB each, i.e. they live on the unsigned 32-bit buffer-offset range above 2^31 bytes) — every image
        against `oracle.pipeline_ref.single_infer_ref` (Marigold/marigold/marigold_pipeline.py:372-478), depth AND normals;
  (ii)  the per-op shapes of that batch whose tensors cross 2^31 bytes, against torch CPU fp32 on the FIRST and the LAST image (images are
        independent, so the reference is computed for those two only): fused nearest-upsample conv 256->256 @384^2->768^2, conv 256->128
        @768^2, GroupNorm(+SiLU) 256 ch @768^2, the VAE mid-block attention on 8 x 9216 tokens of width 512;
  (iii) BASELINE.json configs[4] per-GPU share at full size — GeoWizard, 768x768, fp16: UNet batch 2 per image with cross-domain joint
        self-attention over 2 x 9216 = 18432 keys (GeoWizard/geowizard/models/geowizard_pipeline.py:252-344, attention.py:482-491) against
        `oracle.pipeline_ref.geowizard_infer_ref`, depth and normals.
Tolerances are those of tests/test_fullsize_parity_gpu.py for 16-bit compute against the fp32 oracle: depth max |err| <= 2e-2 of the output
range and mean |err| <= 2e-3, normals mean angle <= 1 degree; single ops 3e-3 .. 6e-3 (max-abs / max-ref) with the mean-abs error printed
next to it.  CPU cost of the oracle on the GPU box's host (32+ cores): ~11 s per 768^2 image (~90 s for the eight), ~25 s GeoWizard."""
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
    torch.set_num_threads(max(n, min(64, os.cpu_count() or 1)))
    yield
    torch.set_num_threads(n)


def _images(n, res, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (n, 3, res, res), generator=g, dtype=torch.int64).float() / 255.0 * 2.0 - 1.0


def _mean_angle(normal, want):
    cos = (TF.normalize(normal.float().cpu(), dim=1) * want).sum(1).clamp(-1, 1)
    return torch.rad2deg(torch.acos(cos)).mean().item()


# ---- (i) configs[1] exactly as benchmarked ------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def marigold_fp16(dev):
    from diffusion_e2e_ft_amd.pipeline import MarigoldPipeline
    from diffusion_e2e_ft_amd.scheduler import DDIMScheduler
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
    ctx = 0.5 * torch.randn((1, 2, 1024), generator=torch.Generator().manual_seed(0))
    pipe = MarigoldPipeline(unet.half().eval(), vae.half().eval(), DDIMScheduler())
    pipe.empty_text_embed = ctx.to(dev, torch.float16)
    return pipe, usd, vsd, ctx


def test_config1_batch8_768_fp16_every_image_against_oracle(dev, marigold_fp16, cpu_threads):
    """batch 8 at 768x768, fp16, one call — the workload of bench.py's default line — image by image against the fp32 CPU oracle"""
    pipe, usd, vsd, ctx = marigold_fp16
    rgb = _images(8, 768, seed=21)
    with torch.no_grad():
        depth = pipe.single_infer(rgb, 1, False, noise="zeros", normals=False).float().cpu()
        normal = pipe.single_infer(rgb, 1, False, noise="zeros", normals=True).float().cpu()
        latent = pipe.predict_latent(rgb).float().cpu()        # the quantity the north star states parity on, in the precision that is benchmarked
    torch.cuda.synchronize()
    assert depth.shape == (8, 1, 768, 768) and normal.shape == (8, 3, 768, 768) and latent.shape == (8, 4, 96, 96)
    worst = dict(e=0.0, mae=0.0, ang=0.0, lat=0.0, lat2=0.0)
    for i in range(8):
        with torch.no_grad():
            want_d, x0 = pipeline_ref.single_infer_ref(usd, config.SD2_UNET, vsd, config.SD_VAE, rgb[i:i + 1], ctx, return_latent=True)
            ang = 0.0
            if i in (0, 3, 7):        # normals (a second decoder pass of the oracle, ~6 s each) on the first, a middle and the last image
                want_n = pipeline_ref.decode_ref(vsd, config.SD_VAE, x0)
                want_n = want_n / (torch.norm(want_n, p=2, dim=1, keepdim=True) + 1e-5)
                ang = _mean_angle(normal[i:i + 1], want_n)
        e = rel_err(depth[i:i + 1], want_d)
        mae = (depth[i:i + 1] - want_d).abs().mean().item()
        # fp16 drift where the north star defines parity: x0 latent against the fp32 oracle's, max-abs / max-ref and relative L2.
        # The fp32 product meets 1e-3 on this quantity at this resolution (test_fullsize_parity_gpu.py::test_config1_768_fp32_latent_and_depth: 7.2e-6 measured);
        # fp16 storage with fp32 accumulation is a STATED looser bar.
        lat = rel_err(latent[i:i + 1], x0)
        lat2 = ((latent[i:i + 1] - x0).norm() / x0.norm()).item()
        print("configs[1] batch 8, image %d: x0 latent max rel err %.3e, rel L2 %.3e; depth max rel err %.3e, mean abs err %.3e; normals mean angle %.3f deg" % (i, lat, lat2, e, mae, ang))
        assert e <= 2e-2 and mae <= 2e-3, (i, e, mae)
        assert ang <= 1.0, (i, ang)
        # bars = 2 x the worst image measured on the round-6 build (profiles/r06a_parity_768_tests.log: max-abs / max 3.24e-3, relative L2 2.61e-3)
        assert lat <= 6.5e-3 and lat2 <= 5.2e-3, (i, lat, lat2)
        worst = dict(e=max(worst["e"], e), mae=max(worst["mae"], mae), ang=max(worst["ang"], ang), lat=max(worst["lat"], lat), lat2=max(worst["lat2"], lat2))
    print("configs[1] batch 8 worst image: x0 latent max rel err %.3e (rel L2 %.3e), depth max rel err %.3e, mean abs err %.3e, normals mean angle %.3f deg"
          % (worst["lat"], worst["lat2"], worst["e"], worst["mae"], worst["ang"]))


# ---- (ii) the tensors of that batch that cross 2^31 bytes ---------------------------------------------------------------------
def _randn_dev(shape, dev, seed, dtype=torch.float16, scale=1.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=dev, dtype=torch.float32) * scale).to(dtype)


def _err_pair(out_nhwc_img, ref_nchw):
    """(max-abs / max-ref, mean-abs / mean-abs-ref) of one image: device NHWC slice vs CPU NCHW reference"""
    o = out_nhwc_img.permute(0, 3, 1, 2).float().cpu()
    return rel_err(o, ref_nchw), ((o - ref_nchw).abs().mean() / ref_nchw.abs().mean().clamp_min(1e-30)).item()


def _conv_b8(dev, Ci, Co, H, W, up_to, seed):
    """conv3x3 pad 1 on a batch-8 NHWC tensor generated on the device; reference on images 0 and 7 only"""
    from diffusion_e2e_ft_amd import ops
    from util import pack_conv_weight
    dtype = torch.float16
    x = _randn_dev((8, H, W, Ci), dev, seed)                              # NHWC
    g = torch.Generator().manual_seed(seed + 1)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).to(dtype).float()
    b = torch.randn(Co, generator=g).to(dtype).float()
    out = ops.conv2d(x, pack_conv_weight(w, dtype, dev), b.to(dtype).to(dev), Co, 3, 3, 1, (1, 1, 1, 1), up_to=up_to)
    torch.cuda.synchronize()
    Ho, Wo = (H, W) if up_to is None else up_to
    assert out.shape == (8, Ho, Wo, Co)
    nbytes_in, nbytes_out = x.numel() * 2, out.numel() * 2
    errs = []
    for i in (0, 7):
        xi = x[i:i + 1].permute(0, 3, 1, 2).float().cpu()
        if up_to is not None:
            xi = TF.interpolate(xi, size=up_to, mode="nearest")
        ref = TF.conv2d(xi.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last), b, padding=1)
        errs.append(_err_pair(out[i:i + 1], ref))
    return errs, nbytes_in, nbytes_out


def test_upsample_conv_256_384_to_768_batch8_last_image(dev, cpu_threads):
    """decoder up-block 2 of the benchmarked batch: [8,384,384,256] -> [8,768,768,256] = 2.42 GB of output (> 2^31 bytes)"""
    errs, nin, nout = _conv_b8(dev, 256, 256, 384, 384, (768, 768), seed=31)
    assert nout > 2 ** 31
    print("upsample conv 256->256 @384->768 B=8 fp16 (out %.2f GB): image 0 max/mean rel err %.2e / %.2e; image 7 %.2e / %.2e" % (nout / 1e9, *errs[0], *errs[1]))
    assert max(e[0] for e in errs) <= 3e-3 and max(e[1] for e in errs) <= 3e-3, errs


def test_conv_256_to_128_at_768_batch8_last_image(dev, cpu_threads):
    """first resnet of the decoder's last up block: the 2.42 GB 256-channel input is read at byte offsets beyond 2^31 for images 7, 8"""
    errs, nin, nout = _conv_b8(dev, 256, 128, 768, 768, None, seed=32)
    assert nin > 2 ** 31
    print("conv 256->128 @768 B=8 fp16 (in %.2f GB): image 0 max/mean rel err %.2e / %.2e; image 7 %.2e / %.2e" % (nin / 1e9, *errs[0], *errs[1]))
    assert max(e[0] for e in errs) <= 3e-3 and max(e[1] for e in errs) <= 3e-3, errs


def test_groupnorm_256_at_768_batch8_last_image(dev, cpu_threads):
    from diffusion_e2e_ft_amd import ops
    dtype = torch.float16
    x = _randn_dev((8, 768, 768, 256), dev, 33, scale=1.5)
    x += _randn_dev((1, 1, 1, 256), dev, 34)
    g = torch.Generator().manual_seed(35)
    ga, be = (1 + 0.3 * torch.randn(256, generator=g)).to(dtype).float(), (0.3 * torch.randn(256, generator=g)).to(dtype).float()
    out = ops.groupnorm(x, ga.to(dtype).to(dev), be.to(dtype).to(dev), 32, 1e-6, True)
    torch.cuda.synchronize()
    assert x.numel() * 2 > 2 ** 31
    errs = []
    for i in (0, 7):
        ref = TF.silu(TF.group_norm(x[i:i + 1].permute(0, 3, 1, 2).float().cpu(), 32, ga, be, 1e-6))
        errs.append(_err_pair(out[i:i + 1], ref))
    print("GroupNorm+SiLU 256 ch @768 B=8 fp16 (2.42 GB): image 0 max/mean rel err %.2e / %.2e; image 7 %.2e / %.2e" % (*errs[0], *errs[1]))
    assert max(e[0] for e in errs) <= 4.5e-3 and max(e[1] for e in errs) <= 3e-3, errs


def test_vae_attention_batch8_9216_tokens_last_image(dev, cpu_threads):
    """AutoencoderKL mid-block attention of the benchmarked batch (unet_2d_blocks.py:589-601): 8 images x 9216 tokens x one 512-wide head
    — the fused d = 512 kernel in 16-bit inference (attn512.hip); before round 3 this was an 8 x 9216 x 9216 score matrix of 1.36 GB."""
    from diffusion_e2e_ft_amd.modules import VaeAttention
    dtype = torch.float16
    C, N = 512, 9216
    m = VaeAttention(C)
    g = torch.Generator().manual_seed(36)
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if n_.startswith("group_norm"):
                p.copy_((1.0 if n_.endswith("weight") else 0.0) + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (C ** -0.5 if p.dim() == 2 else 0.1))
            p.copy_(p.to(dtype).float())
    x = _randn_dev((8, 96, 96, C), dev, 37)
    with torch.no_grad():
        out = m.to(dev, dtype).nhwc(x)
    torch.cuda.synchronize()
    mc = m.float().cpu()
    errs = []
    for i in (0, 7):
        xi = x[i:i + 1].float().cpu()                                       # [1,96,96,C]
        with torch.no_grad():
            n = TF.group_norm(xi.permute(0, 3, 1, 2), 32, mc.group_norm.weight, mc.group_norm.bias, 1e-6).permute(0, 2, 3, 1).reshape(1, N, C)
            qv, kv, vv = (TF.linear(n, getattr(mc, k).weight, getattr(mc, k).bias) for k in ("to_q", "to_k", "to_v"))
            a = TF.scaled_dot_product_attention(qv[:, None], kv[:, None], vv[:, None])[:, 0]
            ref = (TF.linear(a, mc.to_out[0].weight, mc.to_out[0].bias) + xi.reshape(1, N, C)).reshape(1, 96, 96, C).permute(0, 3, 1, 2)
        errs.append(_err_pair(out[i:i + 1], ref))
    print("VAE attention d=512 B=8 N=9216 fp16: image 0 max/mean rel err %.2e / %.2e; image 7 %.2e / %.2e" % (*errs[0], *errs[1]))
    assert max(e[0] for e in errs) <= 6e-3 and max(e[1] for e in errs) <= 3e-3, errs


# ---- (iii) GeoWizard at full size ---------------------------------------------------------------------------------------------
def test_config4_geowizard_768_fp16_against_oracle(dev, cpu_threads):
    """one 768x768 image through DepthNormalEstimationPipeline.single_infer in fp16: UNet batch 2, joint self-attention Nk = 18432"""
    from diffusion_e2e_ft_amd.pipeline import DepthNormalEstimationPipeline
    from diffusion_e2e_ft_amd.scheduler import DDIMScheduler
    from diffusion_e2e_ft_amd.synth import init_synthetic_
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    from diffusion_e2e_ft_amd.vae import AutoencoderKL
    with torch.device(dev):
        unet = UNet2DConditionModel(in_channels=8, cross_attention_dim=768, class_embed_type="projection", projection_class_embeddings_input_dim=10,
                                    joint_attention=True)
        vae = AutoencoderKL()
    init_synthetic_(unet, seed=1234)
    init_synthetic_(vae, seed=4321)
    usd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    vsd = {k: v.detach().float().cpu() for k, v in vae.state_dict().items()}
    assert set(usd) == set(unet_ref.unet_param_shapes(config.GEOWIZARD_UNET))
    rgb = _images(1, 768, seed=41)
    emb = 0.5 * torch.randn((1, 1, 768), generator=torch.Generator().manual_seed(42))
    pipe = DepthNormalEstimationPipeline(unet.half().eval(), vae.half().eval(), DDIMScheduler())
    with torch.no_grad():
        depth, normal = pipe.single_infer(rgb, 1, "indoor", img_embed=emb)
    torch.cuda.synchronize()
    del pipe
    with torch.no_grad():
        want_d, want_n = pipeline_ref.geowizard_infer_ref(usd, config.GEOWIZARD_UNET, vsd, config.SD_VAE, rgb, emb, "indoor")
    e = rel_err(depth.float(), want_d)
    mae = (depth.float().cpu() - want_d).abs().mean().item()
    ang = _mean_angle(normal, TF.normalize(want_n, dim=1))
    print("GeoWizard 768^2 fp16: depth max rel err %.3e, mean abs err %.3e; normals mean angle %.3f deg" % (e, mae, ang))
    assert e <= 2e-2 and mae <= 2e-3, (e, mae)
    assert ang <= 1.0, ang
