"""GPU parity of the CLIP image encoder (diffusion_e2e_ft_amd.clip) and the activation kernel against the CPU oracle
(oracle/clip_ref.py, itself pinned to transformers in tests/test_clip_cpu.py)."""
import pytest
import torch

from oracle import clip_ref
from test_clip_cpu import TINY, tiny_clip_sd
from util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kind", ["quick_gelu", "gelu", "silu"])
@pytest.mark.parametrize("n", [8, 1000, 4099, 257 * 4096])
def test_activation(dev, dtype, kind, n):
    from diffusion_e2e_ft_amd import ops
    x = (3 * torch.randn(n, generator=torch.Generator().manual_seed(n))).to(dtype)
    xf = x.float()
    ref = {"quick_gelu": xf * torch.sigmoid(1.702 * xf), "gelu": torch.nn.functional.gelu(xf), "silu": torch.nn.functional.silu(xf)}[kind]
    y = ops.activation(x.to(dev), kind).cpu().float()
    tol = 2e-6 if dtype == torch.float32 else (1e-3 if dtype == torch.float16 else 8e-3)
    assert (y - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype,image", [(torch.float32, 56), (torch.float16, 56), (torch.bfloat16, 56), (torch.float16, 224)])
def test_clip_vision_tiny(dev, dtype, image):
    from diffusion_e2e_ft_amd.clip import CLIPVisionModelWithProjection
    cfg = dict(TINY, image_size=image)
    sd = tiny_clip_sd(cfg=cfg)                     # image 224 -> 257 tokens: the ragged sequence length of ViT-L/14 on the fused attention kernel
    m = CLIPVisionModelWithProjection(**cfg)
    m.load_state_dict(sd)
    m = m.to(device=dev, dtype=dtype).eval()
    x = torch.randn(2, 3, image, image, generator=torch.Generator().manual_seed(5))
    emb_ref, last_ref = clip_ref.clip_vision_ref(sd, cfg, x)
    out = m(x.to(dev, dtype))
    tol = 1e-3 if dtype == torch.float32 else (2e-2 if dtype == torch.float16 else 8e-2)
    assert tuple(out.image_embeds.shape) == (2, cfg["projection_dim"])
    e1, e2 = rel_err(out.image_embeds.float().cpu(), emb_ref), rel_err(out.last_hidden_state.float().cpu(), last_ref)
    assert e1 <= tol and e2 <= tol, (e1, e2)


def test_clip_against_transformers_on_gpu_box(dev):
    tr = pytest.importorskip("transformers")
    from diffusion_e2e_ft_amd.clip import CLIPVisionModelWithProjection
    sd = tiny_clip_sd()
    ref = tr.CLIPVisionModelWithProjection(tr.CLIPVisionConfig(**TINY)).eval()
    ref.load_state_dict(sd, strict=False)
    m = CLIPVisionModelWithProjection(**TINY)
    m.load_state_dict(ref.state_dict())            # accepts (and ignores) a position_ids buffer if this transformers release has one
    m = m.to(dev).eval()
    x = torch.randn(1, 3, 56, 56, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        want = ref(pixel_values=x).image_embeds
    assert rel_err(m(x.to(dev)).image_embeds.cpu(), want) <= 1e-3


def test_geowizard_pipeline_with_image_encoder(dev):
    """geowizard_pipeline.py:232-248,283-284: the embedding is recomputed from the input image on every call"""
    import golden_cases as gc
    from oracle import config
    from diffusion_e2e_ft_amd.clip import CLIPVisionModelWithProjection, preprocess_for_clip
    from diffusion_e2e_ft_amd.pipeline import DepthNormalEstimationPipeline
    from diffusion_e2e_ft_amd.scheduler import DDIMScheduler
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    from diffusion_e2e_ft_amd.vae import AutoencoderKL
    rgb, ctx = gc.geo_pipe_inputs()
    xdim = ctx.shape[-1]
    unet = UNet2DConditionModel(**config.TINY_GEOWIZARD_UNET)
    unet.load_state_dict(gc.tiny_geo_sd())
    vae = AutoencoderKL(**config.TINY_VAE)
    vae.load_state_dict(gc.tiny_vae_sd())
    cfg = dict(TINY, projection_dim=xdim)
    enc = CLIPVisionModelWithProjection(**cfg)
    enc.load_state_dict(tiny_clip_sd(cfg=cfg))
    pipe = DepthNormalEstimationPipeline(unet.to(dev).eval(), vae.to(dev).eval(), DDIMScheduler(), image_encoder=enc.to(dev).eval())
    emb = pipe.encode_img_embed(rgb.to(dev))
    assert tuple(emb.shape) == (rgb.shape[0], 1, xdim)
    emb_ref, _ = clip_ref.clip_vision_ref(tiny_clip_sd(cfg=cfg), cfg, preprocess_for_clip(rgb, cfg["image_size"]))
    assert rel_err(emb[:, 0].cpu(), emb_ref) <= 1e-3
    d1, n1 = pipe.single_infer(rgb)                 # embedding computed inside
    d2, n2 = pipe.single_infer(rgb, img_embed=emb)  # embedding supplied
    assert torch.equal(d1, d2) and torch.equal(n1, n2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("nq", [2, 5, 77])
def test_softmax_rows_causal(dev, dtype, nq):
    from diffusion_e2e_ft_amd import ops
    heads, epc = 3, ops.epc(dtype)
    lds = ops.round_up(nq, epc)
    s = torch.randn(heads * nq, lds, generator=torch.Generator().manual_seed(nq)).to(dtype)
    sf = s.float()[:, :nq].view(heads, nq, nq)
    mask = torch.ones(nq, nq, dtype=torch.bool).tril()
    ref = torch.softmax((sf * 0.37).masked_fill(~mask, float("-inf")), dim=-1)
    out = ops.softmax_rows_(s.to(dev), nq, 0.37, causal_nq=nq).cpu().float()
    assert (out[:, :nq].view(heads, nq, nq) - ref).abs().max().item() <= (1e-6 if dtype == torch.float32 else 1e-3)
    assert (out[:, nq:] == 0).all() and (out[:, :nq].view(heads, nq, nq)[:, ~mask] == 0).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("padding", ["do_not_pad", "max_length"])
def test_clip_text_tiny(dev, dtype, padding):
    from diffusion_e2e_ft_amd.clip import CLIPTextModel, empty_prompt_ids
    from test_clip_cpu import TINY_TEXT, tiny_text_sd
    sd = tiny_text_sd()
    m = CLIPTextModel(**TINY_TEXT)
    m.load_state_dict(sd)
    m = m.to(device=dev, dtype=dtype).eval()
    ids = empty_prompt_ids(padding)
    want = clip_ref.clip_text_ref(sd, TINY_TEXT, ids)
    got = m(ids)[0]
    assert tuple(got.shape) == tuple(want.shape)
    assert rel_err(got.float().cpu(), want) <= (1e-3 if dtype == torch.float32 else 2e-2)
    assert torch.equal(m(ids, return_dict=False)[0], got)


def test_pipeline_encode_empty_text_with_native_text_encoder(dev):
    """marigold_pipeline.py:356-369 with the text_encoder slot filled by clip.CLIPTextModel and no tokenizer files"""
    import golden_cases as gc
    from oracle import config
    from diffusion_e2e_ft_amd.clip import CLIPTextModel, empty_prompt_ids
    from diffusion_e2e_ft_amd.pipeline import MarigoldPipeline
    from diffusion_e2e_ft_amd.scheduler import DDIMScheduler
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    from diffusion_e2e_ft_amd.vae import AutoencoderKL
    from test_clip_cpu import TINY_TEXT, tiny_text_sd
    xdim = config.TINY_UNET["cross_attention_dim"]
    cfg = dict(TINY_TEXT, hidden_size=xdim)
    txt = CLIPTextModel(**cfg)
    txt.load_state_dict(tiny_text_sd(cfg=cfg))
    unet = UNet2DConditionModel(**config.TINY_UNET)
    unet.load_state_dict(gc.tiny_unet_sd())
    vae = AutoencoderKL(**config.TINY_VAE)
    vae.load_state_dict(gc.tiny_vae_sd())
    pipe = MarigoldPipeline(unet.to(dev).eval(), vae.to(dev).eval(), DDIMScheduler(), text_encoder=txt.to(dev).eval(), tokenizer=None)
    pipe.encode_empty_text()
    assert tuple(pipe.empty_text_embed.shape) == (1, 2, xdim)
    want = clip_ref.clip_text_ref(tiny_text_sd(cfg=cfg), cfg, empty_prompt_ids("do_not_pad"))
    assert rel_err(pipe.empty_text_embed.float().cpu(), want) <= 1e-3
    from oracle import synth
    rgb, _ = synth.synth_inputs(1, 64, 64, 2, xdim, seed=3)
    out = pipe.single_infer(rgb, 1, noise="zeros")     # the embedding produced above drives the UNet's cross-attention
    assert torch.isfinite(out).all() and tuple(out.shape) == (1, 1, 64, 64)


def test_geowizard_pipeline_hip_graph_replay(dev):
    """DepthNormalEstimationPipeline.enable_hip_graphs(): CLIP tower + dual-latent UNet + two decodes replayed from one graph, bit-equal
    to the eager launches, with the embedding computed inside the graph or supplied."""
    import golden_cases as gc
    from oracle import config
    from diffusion_e2e_ft_amd.clip import CLIPVisionModelWithProjection
    from diffusion_e2e_ft_amd.pipeline import DepthNormalEstimationPipeline
    from diffusion_e2e_ft_amd.scheduler import DDIMScheduler
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    from diffusion_e2e_ft_amd.vae import AutoencoderKL
    dtype = torch.float16
    rgb, ctx = gc.geo_pipe_inputs()
    rgb2 = rgb.flip(-1).contiguous()
    xdim = ctx.shape[-1]
    unet = UNet2DConditionModel(**config.TINY_GEOWIZARD_UNET)
    unet.load_state_dict(gc.tiny_geo_sd())
    vae = AutoencoderKL(**config.TINY_VAE)
    vae.load_state_dict(gc.tiny_vae_sd())
    cfg = dict(TINY, projection_dim=xdim)
    enc = CLIPVisionModelWithProjection(**cfg)
    enc.load_state_dict(tiny_clip_sd(cfg=cfg))
    pipe = DepthNormalEstimationPipeline(unet.to(dev, dtype).eval(), vae.to(dev, dtype).eval(), DDIMScheduler(), image_encoder=enc.to(dev, dtype).eval())
    eager = [tuple(t.clone() for t in pipe.single_infer(r)) for r in (rgb, rgb2)]
    eager_emb = tuple(t.clone() for t in pipe.single_infer(rgb, img_embed=ctx))
    pipe.enable_hip_graphs()
    for r, want in ((rgb, eager[0]), (rgb2, eager[1]), (rgb, eager[0])):
        d, n = pipe.single_infer(r)
        assert torch.equal(d, want[0]) and torch.equal(n, want[1])
    d, n = pipe.single_infer(rgb, img_embed=ctx)
    assert torch.equal(d, eager_emb[0]) and torch.equal(n, eager_emb[1])
    assert len(pipe._graphs) == 2


def test_geowizard_pipeline_call(dev):
    """DepthNormalEstimationPipeline.__call__ (geowizard_pipeline.py:88-230): tensor input, resize to processing_res and back,
    ensembling of the (identical, zero-noise) passes, HWC normals"""
    import golden_cases as gc
    from oracle import config
    from diffusion_e2e_ft_amd.clip import CLIPVisionModelWithProjection
    from diffusion_e2e_ft_amd.pipeline import DepthNormalEstimationPipeline
    from diffusion_e2e_ft_amd.scheduler import DDIMScheduler
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    from diffusion_e2e_ft_amd.vae import AutoencoderKL
    import warnings
    rgb, ctx = gc.geo_pipe_inputs()
    xdim = ctx.shape[-1]
    unet = UNet2DConditionModel(**config.TINY_GEOWIZARD_UNET)
    unet.load_state_dict(gc.tiny_geo_sd())
    vae = AutoencoderKL(**config.TINY_VAE)
    vae.load_state_dict(gc.tiny_vae_sd())
    cfg = dict(TINY, projection_dim=xdim)
    enc = CLIPVisionModelWithProjection(**cfg)
    enc.load_state_dict(tiny_clip_sd(cfg=cfg))
    pipe = DepthNormalEstimationPipeline(unet.to(dev).eval(), vae.to(dev).eval(), DDIMScheduler(), image_encoder=enc.to(dev).eval())
    img = ((rgb[0] + 1) / 2 * 255).round()                       # [3, 64, 64] in 0..255
    one = pipe(img, processing_res=0, match_input_res=True)
    assert one.depth_np.shape == (64, 64) and one.normal_np.shape == (64, 64, 3) and one.uncertainty is None
    assert float(one.depth_np.min()) == 0.0 and float(one.depth_np.max()) == 1.0
    d_ref, n_ref = pipe.single_infer((img / 255 * 2 - 1)[None])
    assert abs(float(((one.normal_np ** 2).sum(-1) ** 0.5).mean()) - 1.0) < 1e-3
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ens = pipe(img, ensemble_size=3, batch_size=2, processing_res=0)
    assert torch.allclose(torch.from_numpy(ens.depth_np), torch.from_numpy(one.depth_np), atol=1e-5)   # identical passes -> same ensemble
    assert tuple(ens.uncertainty.shape) == (64, 64) and float(ens.uncertainty.abs().max()) < 1e-6
    big = torch.nn.functional.interpolate(img[None], size=(128, 96), mode="bilinear")[0]
    out = pipe(big, processing_res=64, match_input_res=True)      # processed at 64 x 48, resized back
    assert out.depth_np.shape == (128, 96) and out.normal_np.shape == (128, 96, 3)
    # the original (multi-step, noisy) GeoWizard setting: 3 DDIM steps, 3 ensemble members with different noise
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        multi = pipe(img, denoising_steps=3, ensemble_size=3, batch_size=3, processing_res=0, noise="gaussian")
    assert multi.depth_np.shape == (64, 64) and float(multi.uncertainty.float().max()) > 0     # members differ now
    assert abs(float(((multi.normal_np ** 2).sum(-1) ** 0.5).mean()) - 1.0) < 1e-3
    # one noisy step reproduces the scheduler arithmetic of the Marigold loop: same UNet call, same DDIM step
    g = torch.Generator(device=dev).manual_seed(3)
    d1, n1 = pipe.single_infer((img / 255 * 2 - 1)[None], num_inference_steps=2, noise="gaussian", generator=g)
    g = torch.Generator(device=dev).manual_seed(3)
    d2, n2 = pipe.single_infer((img / 255 * 2 - 1)[None], num_inference_steps=2, noise="gaussian", generator=g)
    assert torch.equal(d1, d2) and torch.equal(n1, n2) and torch.isfinite(d1).all()


@pytest.mark.parametrize("steps", [2, 3])
def test_geowizard_multistep_matches_oracle(dev, steps):
    """the original GeoWizard setting (DDIM over the joint geometry latent, geowizard_pipeline.py:266-343) against the CPU oracle's
    restatement of the loop and of diffusers' DDIM v-prediction step, from the same explicit initial latent"""
    import golden_cases as gc
    from oracle import config, pipeline_ref
    from diffusion_e2e_ft_amd.pipeline import DepthNormalEstimationPipeline
    from diffusion_e2e_ft_amd.scheduler import DDIMScheduler
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    from diffusion_e2e_ft_amd.vae import AutoencoderKL
    rgb, emb = gc.geo_pipe_inputs()
    usd, vsd = gc.tiny_geo_sd(), gc.tiny_vae_sd()
    init = 0.8 * torch.randn(rgb.shape[0], 4, rgb.shape[2] // 8, rgb.shape[3] // 8, generator=torch.Generator().manual_seed(steps))
    want_d, want_n = pipeline_ref.geowizard_multistep_ref(usd, config.TINY_GEOWIZARD_UNET, vsd, config.TINY_VAE, rgb, emb, init, steps)
    unet = UNet2DConditionModel(**config.TINY_GEOWIZARD_UNET)
    unet.load_state_dict(usd)
    vae = AutoencoderKL(**config.TINY_VAE)
    vae.load_state_dict(vsd)
    pipe = DepthNormalEstimationPipeline(unet.to(dev).eval(), vae.to(dev).eval(), DDIMScheduler())
    d, n = pipe.single_infer(rgb, img_embed=emb, num_inference_steps=steps, noise=init)
    assert rel_err(d.cpu(), want_d) <= 2e-3, rel_err(d.cpu(), want_d)
    cos = (n.cpu() * want_n).sum(1)
    assert float(cos.mean()) > 0.999 and float(cos.min()) > 0.9
