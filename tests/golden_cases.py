"""Seeded cases shared by tests/golden/make_golden.py (which stores the oracle's / reference's outputs) and the tests
(which regenerate the same inputs and weights from the seeds)."""
import torch

from oracle import config, unet_ref, vae_ref, pipeline_ref, synth

LR_ITERS = [0, 1, 50, 99, 100, 101, 5000, 10050, 19999, 20000, 30000]


def tiny_unet_sd():
    return synth.synth_state_dict(unet_ref.unet_param_shapes(config.TINY_UNET), seed=1234)


def tiny_vae_sd():
    return synth.synth_state_dict(vae_ref.vae_param_shapes(config.TINY_VAE), seed=4321)


def tiny_geo_sd():
    return synth.synth_state_dict(unet_ref.unet_param_shapes(config.TINY_GEOWIZARD_UNET), seed=99)


def unet_inputs(hw, batch=2, ctx_len=2, seed=7, xdim=128):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, 8, *hw, generator=g), 0.5 * torch.randn(batch, ctx_len, xdim, generator=g)


def vae_inputs():
    g = torch.Generator().manual_seed(10)
    return torch.rand(2, 3, 40, 56, generator=g) * 2 - 1, torch.randn(2, 4, 5, 7, generator=g)


def geo_unet_inputs(Bh=2):
    cfg = config.TINY_GEOWIZARD_UNET
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2 * Bh, 8, 16, 16, generator=g)
    ctx = 0.5 * torch.randn(2 * Bh, 1, cfg["cross_attention_dim"], generator=g)
    return x, ctx, pipeline_ref.geowizard_class_embedding(Bh, "outdoor")


def geo_pipe_inputs():
    cfg = config.TINY_GEOWIZARD_UNET
    rgb, _ = synth.synth_inputs(2, 64, 64, 1, cfg["cross_attention_dim"], seed=5)
    emb = 0.5 * torch.randn(2, 1, cfg["cross_attention_dim"], generator=torch.Generator().manual_seed(6))
    return rgb, emb


def ssi_inputs():
    g = torch.Generator().manual_seed(61)
    B, H, W = 3, 40, 56
    tgt = torch.rand(B, 1, H, W, generator=g) * 2 - 1
    pred = 0.6 * tgt + 0.2 + 0.05 * torch.randn(B, 1, H, W, generator=g)
    mask = torch.rand(B, 1, H, W, generator=g) > 0.05
    mask[2] = False
    return pred, tgt, mask


def angular_inputs():
    g = torch.Generator().manual_seed(62)
    B, H, W = 3, 40, 56
    n = torch.nn.functional.normalize(torch.randn(B, 3, H, W, generator=g), dim=1)
    nt = torch.nn.functional.normalize(n + 0.3 * torch.randn(B, 3, H, W, generator=g), dim=1)
    mask = torch.rand(B, 1, H, W, generator=g) > 0.05
    return n, nt, mask


def _unet(hw, **kw):
    x, ctx = unet_inputs(hw, **kw)
    return {"out": unet_ref.unet_forward(tiny_unet_sd(), config.TINY_UNET, x, 999, ctx)}


def _unet77():
    x, ctx = unet_inputs((8, 8), batch=3, ctx_len=77, seed=8)
    return {"out": unet_ref.unet_forward(tiny_unet_sd(), config.TINY_UNET, x, torch.full((3,), 999), ctx)}


def _geo_unet():
    x, ctx, cls = geo_unet_inputs()
    return {"out": unet_ref.unet_forward(tiny_geo_sd(), config.TINY_GEOWIZARD_UNET, x, 999, ctx, class_labels=cls)}


def _vae():
    rgb, z = vae_inputs()
    sd = tiny_vae_sd()
    return {"moments": vae_ref.quant_conv(sd, vae_ref.encoder_forward(sd, config.TINY_VAE, rgb)),
            "dec": vae_ref.decoder_forward(sd, config.TINY_VAE, vae_ref.post_quant_conv(sd, z))}


def _pipe():
    rgb, ctx = synth.synth_inputs(2, 64, 96, 2, 128, seed=3)
    usd, vsd = tiny_unet_sd(), tiny_vae_sd()
    d, x0 = pipeline_ref.single_infer_ref(usd, config.TINY_UNET, vsd, config.TINY_VAE, rgb, ctx, normals=False, return_latent=True)
    n = pipeline_ref.single_infer_ref(usd, config.TINY_UNET, vsd, config.TINY_VAE, rgb, ctx, normals=True)
    return {"depth": d, "x0": x0, "normal": n}


def _geo_pipe():
    rgb, emb = geo_pipe_inputs()
    d, n, x0 = pipeline_ref.geowizard_infer_ref(tiny_geo_sd(), config.TINY_GEOWIZARD_UNET, tiny_vae_sd(), config.TINY_VAE, rgb, emb,
                                                "indoor", return_latent=True)
    return {"depth": d, "normal": n, "x0": x0}


MODEL_CASES = {
    "unet_16x16": lambda: _unet((16, 16)),
    "unet_20x12": lambda: _unet((20, 12)),
    "unet_ctx77": _unet77,
    "geo_unet": _geo_unet,
    "vae": _vae,
    "pipe": _pipe,
    "geo_pipe": _geo_pipe,
}


def train_batch(seed=21, B=2, H=64, W=64):
    """synthetic micro-batch of the E2E-FT step: rgb in [-1,1], metric depth target in [-1,1], unit normals, 5 % invalid"""
    rgb, ctx = synth.synth_inputs(B, H, W, 2, 128, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    metric = torch.stack([(0.6 * xx * (b + 1) / B + 0.3 * yy).clamp(-1, 1) for b in range(B)])[:, None] + 0.05 * torch.randn(B, 1, H, W, generator=g)
    normals = torch.nn.functional.normalize(torch.randn(B, 3, H, W, generator=g) + torch.tensor([0.0, 0.0, 2.0]).view(1, 3, 1, 1), dim=1)
    mask = torch.rand(B, 1, H, W, generator=g) > 0.05
    return {"rgb": rgb, "metric": metric.clamp(-1, 1), "normals": normals, "val_mask": mask}, ctx[:1]


TRAIN_FULL_GRADS = ["conv_in.weight", "conv_in.bias", "time_embedding.linear_1.weight", "down_blocks.0.resnets.0.time_emb_proj.weight",
                    "down_blocks.0.resnets.0.norm1.weight", "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight",
                    "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight", "down_blocks.0.attentions.0.transformer_blocks.0.norm2.bias",
                    "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.weight", "down_blocks.0.downsamplers.0.conv.weight",
                    "mid_block.resnets.0.conv1.weight", "up_blocks.0.resnets.0.conv1.weight", "up_blocks.0.resnets.0.conv_shortcut.weight",
                    "up_blocks.0.upsamplers.0.conv.weight", "conv_norm_out.bias", "conv_out.weight", "conv_out.bias"]


def train_grads(modality):
    """loss and UNet parameter gradients of one micro-step through the CPU oracle (torch autograd over oracle/*_ref.py)"""
    batch, text = train_batch()
    usd = {k: v.clone().requires_grad_(True) for k, v in tiny_unet_sd().items()}
    loss, est = pipeline_ref.train_forward_ref(usd, config.TINY_UNET, tiny_vae_sd(), config.TINY_VAE, batch, text, modality)
    loss.backward()
    out = {"loss": loss.detach(), "estimate": est.detach(),
           "grad_norms": {k: float(v.grad.norm()) for k, v in usd.items()}}
    keys = [k for k in TRAIN_FULL_GRADS if k in usd]
    assert len(keys) >= 12, [k for k in TRAIN_FULL_GRADS if k not in usd]
    # large tensors are stored as a strided sample of the flattened gradient (GRAD_STRIDE) to keep the fixture small
    out["grads"] = {k: sample_grad(usd[k].grad) for k in keys}
    out["estimate"] = out["estimate"][:, :, ::4, ::4].clone()
    return out


GRAD_STRIDE = 61


def sample_grad(g):
    f = g.detach().reshape(-1)
    return f.clone() if f.numel() <= 4096 else f[::GRAD_STRIDE].clone()


def geo_train_inputs():
    batch, _ = train_batch(seed=31)
    emb = 0.5 * torch.randn(2, 1, config.TINY_GEOWIZARD_UNET["cross_attention_dim"], generator=torch.Generator().manual_seed(32))
    return batch, emb


def geo_train_grads():
    """GeoWizard E2E-FT micro-step (joint attention, class embedding, 0.5 SSI + angular) through autograd over the CPU oracle"""
    batch, emb = geo_train_inputs()
    usd = {k: v.clone().requires_grad_(True) for k, v in tiny_geo_sd().items()}
    loss, ssi, ang = pipeline_ref.geowizard_train_forward_ref(usd, config.TINY_GEOWIZARD_UNET, tiny_vae_sd(), config.TINY_VAE, batch, emb, "indoor")
    loss.backward()
    keys = [k for k in TRAIN_FULL_GRADS if k in usd] + ["class_embedding.linear_1.weight"]
    return {"loss": loss.detach(), "ssi": ssi.detach(), "angular": ang.detach(),
            "grad_norms": {k: float(v.grad.norm()) for k, v in usd.items()}, "grads": {k: sample_grad(usd[k].grad) for k in keys}}


# ---- test-time ensembling (oracle/ensemble_ref.py, tests/golden/make_ensemble_golden.py) -------------------------------------
def ensemble_depth_stack(n=10, H=48, W=64, seed=31):
    """N affine-distorted, noisy copies of one smooth depth map (what the N diffusion samples of one image look like)"""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    base = 0.3 + 0.5 * yy + 0.2 * torch.sin(6 * xx) * yy
    rows = [(0.5 + torch.rand(1, generator=g)) * base + 0.3 * torch.randn(1, generator=g) + 0.02 * torch.randn(H, W, generator=g)
            for _ in range(n)]
    return torch.stack(rows).float()


def ensemble_normal_stack(n=6, H=40, W=56, seed=33):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    base = torch.stack([0.6 * xx, 0.4 * yy + 0.2 * torch.sin(3 * xx), 0.3 + 0.7 * (1 - 0.5 * (xx ** 2 + yy ** 2))])
    return torch.stack([1.3 * base + 0.15 * torch.randn(3, H, W, generator=g) for _ in range(n)]).float()


ENSEMBLE_DEPTH_CASES = {
    "median10": dict(stack=dict(n=10), kw={}),
    "median7_odd": dict(stack=dict(n=7, H=33, W=47, seed=35), kw={}),
    "mean5": dict(stack=dict(n=5, seed=36), kw=dict(reduction="mean")),
    "median4_maxres": dict(stack=dict(n=4, H=40, W=72, seed=37), kw=dict(max_res=32, max_iter=5)),
    "median2": dict(stack=dict(n=2, seed=38), kw=dict(regularizer_strength=0.1)),
}
