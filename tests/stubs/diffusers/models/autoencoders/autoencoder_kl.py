"""diffusers AutoencoderKL (0.30.2) — the four sub-modules the reference calls one by one (marigold_pipeline.py:493-494,515-516;
training/train.py:234-235,241-242) and `.config.scaling_factor`."""
from torch import nn

from ...configuration_utils import ConfigMixin, register_to_config
from ..modeling_utils import ModelMixin
from .vae import Decoder, Encoder


class AutoencoderKL(ModelMixin, ConfigMixin):
    @register_to_config
    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",), up_block_types=("UpDecoderBlock2D",),
                 block_out_channels=(64,), layers_per_block=1, act_fn="silu", latent_channels=4, norm_num_groups=32, sample_size=32,
                 scaling_factor=0.18215, shift_factor=None, latents_mean=None, latents_std=None, force_upcast=True,
                 use_quant_conv=True, use_post_quant_conv=True, mid_block_add_attention=True):
        super().__init__()
        self.encoder = Encoder(in_channels=in_channels, out_channels=latent_channels, down_block_types=down_block_types,
                               block_out_channels=block_out_channels, layers_per_block=layers_per_block, act_fn=act_fn,
                               norm_num_groups=norm_num_groups, double_z=True, mid_block_add_attention=mid_block_add_attention)
        self.decoder = Decoder(in_channels=latent_channels, out_channels=out_channels, up_block_types=up_block_types,
                               block_out_channels=block_out_channels, layers_per_block=layers_per_block, norm_num_groups=norm_num_groups,
                               act_fn=act_fn, mid_block_add_attention=mid_block_add_attention)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1) if use_quant_conv else None
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1) if use_post_quant_conv else None
