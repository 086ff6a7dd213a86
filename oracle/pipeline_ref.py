"""CPU restatement of the pipeline / training-step glue around the UNet and VAE.  TEST INFRASTRUCTURE ONLY.

  single_infer_ref     Marigold/marigold/marigold_pipeline.py:372-478 (+ encode_rgb :481-498, decode_depth :501-519,
                       decode_normal :522-538) with the E2E-FT defaults (1 step, zeros latent, trailing => t=999;
                       Marigold/run.py:80-89,145-162)
  geowizard_infer_ref  GeoWizard/geowizard/models/geowizard_pipeline.py:252-344, batched as in
                       GeoWizard/geowizard/training/train_depth_normal.py:687-704
  train_forward_ref    training/train.py:470-556 (forward half of the E2E-FT step)
  scheduler constants  diffusers DDIMScheduler (scaled_linear betas, v_prediction) — SURVEY.md Appendix A.3
"""
import torch

from . import unet_ref, vae_ref
from .losses_ref import ssi_loss_ref, angular_loss_ref

SCALING = 0.18215  # marigold_pipeline.py:134-135; vae.config.scaling_factor


def alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def trailing_timesteps(n, num_train_timesteps=1000):
    import numpy as np
    return (np.round(np.arange(num_train_timesteps, 0, -num_train_timesteps / n)) - 1).astype(np.int64)


def v_to_x0(v, x_t, t):
    """DDIM v-prediction: x0 = sqrt(abar_t) x_t - sqrt(1 - abar_t) v  (train.py:509-512; scheduling_ddim.py step)."""
    ac = alphas_cumprod()[t]
    return ac ** 0.5 * x_t - (1 - ac) ** 0.5 * v


def encode_rgb_ref(vae_sd, vae_cfg, rgb):
    h = vae_ref.encoder_forward(vae_sd, vae_cfg, rgb)
    moments = vae_ref.quant_conv(vae_sd, h)
    mean, _ = torch.chunk(moments, 2, dim=1)
    return mean * SCALING


def decode_ref(vae_sd, vae_cfg, latent):
    z = vae_ref.post_quant_conv(vae_sd, latent / SCALING)
    return vae_ref.decoder_forward(vae_sd, vae_cfg, z)


def single_infer_ref(unet_sd, unet_cfg, vae_sd, vae_cfg, rgb, text_embed, normals=False, return_latent=False):
    """rgb [B,3,H,W] in [-1,1]; text_embed [1,L,X].  1 step, zeros latent, t = 999."""
    rgb_latent = encode_rgb_ref(vae_sd, vae_cfg, rgb)
    latent = torch.zeros_like(rgb_latent)
    t = int(trailing_timesteps(1)[0])
    ctx = text_embed.repeat(rgb_latent.shape[0], 1, 1)
    v = unet_ref.unet_forward(unet_sd, unet_cfg, torch.cat([rgb_latent, latent], dim=1), t, ctx)
    x0 = v_to_x0(v, latent, t)
    dec = decode_ref(vae_sd, vae_cfg, x0)
    if normals:
        out = dec / (torch.norm(dec, p=2, dim=1, keepdim=True) + 1e-5)
    else:
        out = (torch.clip(dec.mean(dim=1, keepdim=True), -1.0, 1.0) + 1.0) / 2.0
    return (out, x0) if return_latent else out


def geowizard_class_embedding(batch, domain, dtype=torch.float32):
    """geowizard_pipeline.py:291-302, batched layout [depth rows (B); normal rows (B)] (train_depth_normal.py:687-688)."""
    geo = torch.tensor([[0.0, 1.0], [1.0, 0.0]], dtype=dtype).repeat_interleave(batch, 0)
    dom = {"indoor": [1.0, 0.0, 0.0], "outdoor": [0.0, 1.0, 0.0], "object": [0.0, 0.0, 1.0]}[domain]
    dom = torch.tensor([dom], dtype=dtype).repeat(2 * batch, 1)
    return torch.cat([torch.sin(geo), torch.cos(geo), torch.sin(dom), torch.cos(dom)], dim=-1)


def geowizard_infer_ref(unet_sd, unet_cfg, vae_sd, vae_cfg, rgb, img_embed, domain="indoor", return_latent=False):
    """rgb [B,3,H,W]; img_embed [B,1,X] (CLIP image embedding per image).  Returns depth [B,1,H,W], normal [B,3,H,W]."""
    B = rgb.shape[0]
    rgb_latent = encode_rgb_ref(vae_sd, vae_cfg, rgb)
    geo = torch.zeros_like(rgb_latent).repeat(2, 1, 1, 1)
    t = int(trailing_timesteps(1)[0])
    ctx = img_embed.repeat(2, 1, 1)
    cls = geowizard_class_embedding(B, domain, rgb.dtype)
    v = unet_ref.unet_forward(unet_sd, unet_cfg, torch.cat([rgb_latent.repeat(2, 1, 1, 1), geo], dim=1), t, ctx, class_labels=cls)
    x0 = v_to_x0(v, geo, t)
    dd = decode_ref(vae_sd, vae_cfg, x0[:B])
    depth = (torch.clip(dd.mean(dim=1, keepdim=True), -1.0, 1.0) + 1.0) / 2.0
    dn = decode_ref(vae_sd, vae_cfg, x0[B:])
    normal = -(dn / (torch.norm(dn, p=2, dim=1, keepdim=True) + 1e-5))
    return (depth, normal, x0) if return_latent else (depth, normal)


def ddim_step_ref(v, t, x_t, prev_t):
    """diffusers DDIMScheduler.step for v_prediction, eta = 0, clip_sample False (scheduling_ddim.py): returns (prev_sample, x0).
    prev_t < 0 uses final_alpha_cumprod = alphas_cumprod[0] (set_alpha_to_one False)."""
    ac = alphas_cumprod()
    a_t = ac[t]
    a_prev = ac[prev_t] if prev_t >= 0 else ac[0]
    x0 = a_t ** 0.5 * x_t - (1 - a_t) ** 0.5 * v
    eps = a_t ** 0.5 * v + (1 - a_t) ** 0.5 * x_t
    return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps, x0


def marigold_multistep_ref(unet_sd, unet_cfg, vae_sd, vae_cfg, rgb, text_embed, init_latent, steps, normals=False):
    """marigold_pipeline.py:372-478 for the multi-step checkpoints: DDIM on the trailing schedule from an explicit initial latent, the
    last step returning x0 (:462-465)"""
    rgb_latent = encode_rgb_ref(vae_sd, vae_cfg, rgb)
    latent = init_latent
    ctx = text_embed.repeat(rgb_latent.shape[0], 1, 1)
    ts = [int(t) for t in trailing_timesteps(steps)]
    for i, t in enumerate(ts):
        v = unet_ref.unet_forward(unet_sd, unet_cfg, torch.cat([rgb_latent, latent], dim=1), t, ctx)
        prev, x0 = ddim_step_ref(v, t, latent, t - 1000 // steps)
        latent = x0 if i == steps - 1 else prev
    dec = decode_ref(vae_sd, vae_cfg, latent)
    if normals:
        return dec / (torch.norm(dec, p=2, dim=1, keepdim=True) + 1e-5)
    return (torch.clip(dec.mean(dim=1, keepdim=True), -1.0, 1.0) + 1.0) / 2.0


def geowizard_multistep_ref(unet_sd, unet_cfg, vae_sd, vae_cfg, rgb, img_embed, init_latent, steps, domain="indoor"):
    """geowizard_pipeline.py:266-343 with an explicit initial geometry latent [B,4,h,w] (shared by the depth and the normal row, :271),
    `steps` DDIM steps on the trailing schedule, the last step returning x0 (:335-336)."""
    B = rgb.shape[0]
    rgb_latent = encode_rgb_ref(vae_sd, vae_cfg, rgb).repeat(2, 1, 1, 1)
    geo = init_latent.repeat(2, 1, 1, 1)
    ctx = img_embed.repeat(2, 1, 1)
    cls = geowizard_class_embedding(B, domain, rgb.dtype)
    ts = [int(t) for t in trailing_timesteps(steps)]
    for i, t in enumerate(ts):
        v = unet_ref.unet_forward(unet_sd, unet_cfg, torch.cat([rgb_latent, geo], dim=1), t, ctx, class_labels=cls)
        prev, x0 = ddim_step_ref(v, t, geo, t - 1000 // steps)
        geo = x0 if i == steps - 1 else prev
    dd = decode_ref(vae_sd, vae_cfg, geo[:B])
    depth = (torch.clip(dd.mean(dim=1, keepdim=True), -1.0, 1.0) + 1.0) / 2.0
    dn = decode_ref(vae_sd, vae_cfg, geo[B:])
    return depth, -(dn / (torch.norm(dn, p=2, dim=1, keepdim=True) + 1e-5))


def train_forward_ref(unet_sd, unet_cfg, vae_sd, vae_cfg, batch, text_embed, modality="depth"):
    """Forward half of training/train.py:470-556; returns (loss, current_estimate)."""
    rgb_latents = encode_rgb_ref(vae_sd, vae_cfg, batch["rgb"])
    noisy = torch.zeros_like(rgb_latents)
    t = 999
    ctx = text_embed.repeat(rgb_latents.shape[0], 1, 1)
    v = unet_ref.unet_forward(unet_sd, unet_cfg, torch.cat([rgb_latents, noisy], dim=1), t, ctx)
    x0 = v_to_x0(v, noisy, t)
    est = decode_ref(vae_sd, vae_cfg, x0)
    mask = batch["val_mask"].bool()
    if modality == "depth":
        est = torch.clamp(est.mean(dim=1, keepdim=True), -1, 1)
        loss = ssi_loss_ref(est, batch["metric"], mask)
    else:
        est = torch.clamp(est / (torch.norm(est, p=2, dim=1, keepdim=True) + 1e-5), -1, 1)
        loss = angular_loss_ref(est, batch["normals"], mask)
    return loss, est


def geowizard_train_forward_ref(unet_sd, unet_cfg, vae_sd, vae_cfg, batch, img_embed, domain="indoor"):
    """Forward half of GeoWizard/geowizard/training/train_depth_normal.py:597-768 with --e2e_ft and zeros noise: doubled batch
    [depth rows; normal rows], joint self-attention, class embedding, 0.5 * SSI + 1.0 * angular on inverted normals.
    Returns (loss, ssi, angular)."""
    rgb_latents = encode_rgb_ref(vae_sd, vae_cfg, batch["rgb"])
    B = rgb_latents.shape[0]
    noisy = torch.zeros_like(rgb_latents).repeat(2, 1, 1, 1)
    t = 999
    ctx = img_embed.repeat(2, 1, 1)
    cls = geowizard_class_embedding(B, domain, rgb_latents.dtype)
    v = unet_ref.unet_forward(unet_sd, unet_cfg, torch.cat([rgb_latents.repeat(2, 1, 1, 1), noisy], dim=1), torch.full((2 * B,), t), ctx, class_labels=cls)
    x0 = v_to_x0(v, noisy, t)
    est = decode_ref(vae_sd, vae_cfg, x0)
    d_est, n_est = torch.chunk(est, 2, dim=0)
    d_est = torch.clamp(d_est.mean(dim=1, keepdim=True), -1, 1)
    n_est = torch.clamp(n_est / (torch.norm(n_est, p=2, dim=1, keepdim=True) + 1e-5), -1, 1)
    mask = batch["val_mask"].bool()
    ssi = ssi_loss_ref(d_est, batch["metric"], mask)
    ang = angular_loss_ref(n_est, batch["normals"] * -1, mask)
    return 0.5 * ssi + 1.0 * ang, ssi, ang
