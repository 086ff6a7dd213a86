"""Model configurations for the oracle (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py).

None of these config.json files are in the reference tree (they come from the HF hub: prs-eth/marigold-v1-0,
stabilityai/stable-diffusion-2, lemonaddie/geowizard — /root/reference/training/scripts/train_marigold_e2e_ft_depth.sh:4,
train_geowizard_e2e_ft.sh:4).  Values are the public SD-v2 / SD-VAE definitions, cross-checked by parameter count
(865,910,724 UNet params with in_channels=4; 83,653,863 VAE params)."""

SD2_UNET = dict(
    in_channels=8,  # 4 after hub load; 8 after replace_unet_conv_in (training/util/unet_prep.py:6-20)
    out_channels=4,
    block_out_channels=(320, 640, 1280, 1280),
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    layers_per_block=2,
    attention_head_dim=(5, 10, 20, 20),  # diffusers quirk: this is the NUMBER of heads -> head dim 64
    cross_attention_dim=1024,
    norm_num_groups=32,
    norm_eps=1e-5,
    use_linear_projection=True,
    flip_sin_to_cos=True,
    freq_shift=0,
    class_embed_type=None,
    projection_class_embeddings_input_dim=None,
    joint_attention=False,
    sample_size=96,
)

# GeoWizard variant: CLIP-image context (768), 10-dim projection class embedding, cross-domain joint self-attention
# (/root/reference/GeoWizard/geowizard/models/geowizard_pipeline.py:286-302, attention.py:416-513).
GEOWIZARD_UNET = dict(SD2_UNET, cross_attention_dim=768, class_embed_type="projection",
                      projection_class_embeddings_input_dim=10, joint_attention=True)

SD_VAE = dict(
    in_channels=3,
    out_channels=3,
    latent_channels=4,
    block_out_channels=(128, 256, 512, 512),
    layers_per_block=2,
    norm_num_groups=32,
    scaling_factor=0.18215,
)

# Small configs with the same topology (head dim 64, 32 groups) for tests that must run in seconds on CPU.
TINY_UNET = dict(SD2_UNET, block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128,
                 sample_size=16)
TINY_GEOWIZARD_UNET = dict(GEOWIZARD_UNET, block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4),
                           cross_attention_dim=96, sample_size=16)
TINY_VAE = dict(SD_VAE, block_out_channels=(32, 64, 128, 128))
