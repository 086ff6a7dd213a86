"""diffusers.models.autoencoders.vae (0.30.2) Encoder / Decoder top-level wiring.  NOT in the reference tree; the BLOCKS it is made of
(`DownEncoderBlock2D`, `UpDecoderBlock2D`, `UNetMidBlock2D`, `get_down_block`, `get_up_block`) ARE — the reference vendors them in
GeoWizard/geowizard/models/unet_2d_blocks.py:509-631,1276-1333,2484-2541 — and are taken from there, so a VAE built from this stub runs
the reference's own block code."""
import torch
from torch import nn


def _blocks():
    from geowizard.models import unet_2d_blocks      # the reference's vendored twin of diffusers/models/unets/unet_2d_blocks.py
    return unet_2d_blocks


class Encoder(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",), block_out_channels=(64,), layers_per_block=2,
                 norm_num_groups=32, act_fn="silu", double_z=True, mid_block_add_attention=True):
        super().__init__()
        b = _blocks()
        self.layers_per_block = layers_per_block
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], kernel_size=3, stride=1, padding=1)
        self.down_blocks = nn.ModuleList([])
        output_channel = block_out_channels[0]
        for i, down_block_type in enumerate(down_block_types):
            input_channel = output_channel
            output_channel = block_out_channels[i]
            is_final_block = i == len(block_out_channels) - 1
            self.down_blocks.append(b.get_down_block(
                down_block_type, num_layers=self.layers_per_block, in_channels=input_channel, out_channels=output_channel,
                add_downsample=not is_final_block, resnet_eps=1e-6, downsample_padding=0, resnet_act_fn=act_fn,
                resnet_groups=norm_num_groups, attention_head_dim=output_channel, temb_channels=None))
        self.mid_block = b.UNetMidBlock2D(
            in_channels=block_out_channels[-1], resnet_eps=1e-6, resnet_act_fn=act_fn, output_scale_factor=1,
            resnet_time_scale_shift="default", attention_head_dim=block_out_channels[-1], resnet_groups=norm_num_groups,
            temb_channels=None, add_attention=mid_block_add_attention)
        self.conv_norm_out = nn.GroupNorm(num_channels=block_out_channels[-1], num_groups=norm_num_groups, eps=1e-6)
        self.conv_act = nn.SiLU()
        conv_out_channels = 2 * out_channels if double_z else out_channels
        self.conv_out = nn.Conv2d(block_out_channels[-1], conv_out_channels, 3, padding=1)

    def forward(self, sample):
        sample = self.conv_in(sample)
        for down_block in self.down_blocks:
            sample = down_block(sample)
        sample = self.mid_block(sample)
        sample = self.conv_norm_out(sample)
        sample = self.conv_act(sample)
        return self.conv_out(sample)


class Decoder(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, up_block_types=("UpDecoderBlock2D",), block_out_channels=(64,), layers_per_block=2,
                 norm_num_groups=32, act_fn="silu", norm_type="group", mid_block_add_attention=True):
        super().__init__()
        assert norm_type == "group"
        b = _blocks()
        self.layers_per_block = layers_per_block
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[-1], kernel_size=3, stride=1, padding=1)
        self.up_blocks = nn.ModuleList([])
        temb_channels = None
        self.mid_block = b.UNetMidBlock2D(
            in_channels=block_out_channels[-1], resnet_eps=1e-6, resnet_act_fn=act_fn, output_scale_factor=1,
            resnet_time_scale_shift="default", attention_head_dim=block_out_channels[-1], resnet_groups=norm_num_groups,
            temb_channels=temb_channels, add_attention=mid_block_add_attention)
        reversed_block_out_channels = list(reversed(block_out_channels))
        output_channel = reversed_block_out_channels[0]
        for i, up_block_type in enumerate(up_block_types):
            prev_output_channel = output_channel
            output_channel = reversed_block_out_channels[i]
            is_final_block = i == len(block_out_channels) - 1
            self.up_blocks.append(b.get_up_block(
                up_block_type, num_layers=self.layers_per_block + 1, in_channels=prev_output_channel, out_channels=output_channel,
                prev_output_channel=None, add_upsample=not is_final_block, resnet_eps=1e-6, resnet_act_fn=act_fn,
                resnet_groups=norm_num_groups, attention_head_dim=output_channel, temb_channels=temb_channels,
                resnet_time_scale_shift=norm_type))
            prev_output_channel = output_channel
        self.conv_norm_out = nn.GroupNorm(num_channels=block_out_channels[0], num_groups=norm_num_groups, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, 3, padding=1)

    def forward(self, sample, latent_embeds=None):
        sample = self.conv_in(sample)
        upscale_dtype = next(iter(self.up_blocks.parameters())).dtype
        sample = self.mid_block(sample, latent_embeds)
        sample = sample.to(upscale_dtype)
        for up_block in self.up_blocks:
            sample = up_block(sample, latent_embeds)
        sample = self.conv_norm_out(sample)
        sample = self.conv_act(sample)
        return self.conv_out(sample)
