"""SD `AutoencoderKL` on libe2eft with the attribute surface the reference uses: four separately
callable sub-modules `.encoder(x)`, `.quant_conv(h)`, `.post_quant_conv(z)`, `.decoder(z)` on logical-NCHW tensors
(Marigold/marigold/marigold_pipeline.py:493-494,515-516; training/train.py:234-235,241-242) and
`.config.scaling_factor` (train.py:474,528).  Block structure: diffusers 0.30.2 Encoder/Decoder (twins at
GeoWizard/geowizard/models/unet_2d_blocks.py:509-631,1276-1333,2484-2541)."""
import json
import os

import torch
from torch import nn

from . import ops
from .modules import (Conv2d, GroupNorm, ResnetBlock2D, Downsample2D, Upsample2D, VaeAttention, checkpointed, conv_nhwc, to_nhwc, to_nchw_view)
from .unet import Config

SD_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                     layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)


class _VaeMid(nn.Module):
    def __init__(self, c, g):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, None, g, 1e-6), ResnetBlock2D(c, c, None, g, 1e-6)])
        self.attentions = nn.ModuleList([VaeAttention(c, g)])

    def nhwc(self, h):
        h = self.resnets[0].nhwc(h)
        h = self.attentions[0].nhwc(h)
        return self.resnets[1].nhwc(h)


class _EncBlock(nn.Module):
    def __init__(self, in_c, out_c, layers, g, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_c if j == 0 else out_c, out_c, None, g, 1e-6) for j in range(layers)])
        if add_down:
            self.downsamplers = nn.ModuleList([Downsample2D(out_c, padding=0)])
        self.add_down = add_down

    def nhwc(self, h):
        for r in self.resnets:
            h = r.nhwc(h)
        return self.downsamplers[0].nhwc(h) if self.add_down else h


class _DecBlock(nn.Module):
    def __init__(self, in_c, out_c, layers, g, add_up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_c if j == 0 else out_c, out_c, None, g, 1e-6) for j in range(layers)])
        if add_up:
            self.upsamplers = nn.ModuleList([Upsample2D(out_c)])
        self.add_up = add_up
        self.gradient_checkpointing = False

    def nhwc(self, h):
        for r in self.resnets:
            h = checkpointed(self.gradient_checkpointing, r.nhwc, h)
        return self.upsamplers[0].nhwc(h) if self.add_up else h


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g, L = cfg.block_out_channels, cfg.norm_num_groups, cfg.layers_per_block
        self.conv_in = Conv2d(cfg.in_channels, boc[0], 3, 1, 1)
        blocks, out_c = [], boc[0]
        for i in range(len(boc)):
            in_c, out_c = out_c, boc[i]
            blocks.append(_EncBlock(in_c, out_c, L, g, i != len(boc) - 1))
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _VaeMid(boc[-1], g)
        self.conv_norm_out = GroupNorm(g, boc[-1], eps=1e-6, affine=True)
        self.conv_out = Conv2d(boc[-1], 2 * cfg.latent_channels, 3, 1, 1)

    def nhwc(self, x):
        h = conv_nhwc(self.conv_in, x)
        for b in self.down_blocks:
            h = b.nhwc(h)
        h = self.mid_block.nhwc(h)
        return conv_nhwc(self.conv_out, h, norm=(self.conv_norm_out, True))     # (inference: the norm is applied inside the convolution where the library can)

    @ops.device_scoped
    def forward(self, x):
        return to_nchw_view(self.nhwc(to_nhwc(x)))


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g, L = cfg.block_out_channels, cfg.norm_num_groups, cfg.layers_per_block
        rev = list(reversed(boc))
        self.conv_in = Conv2d(cfg.latent_channels, rev[0], 3, 1, 1)
        self.mid_block = _VaeMid(rev[0], g)
        blocks, out_c = [], rev[0]
        for i in range(len(boc)):
            prev, out_c = out_c, rev[i]
            blocks.append(_DecBlock(prev, out_c, L + 1, g, i != len(boc) - 1))
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = GroupNorm(g, rev[-1], eps=1e-6, affine=True)
        self.conv_out = Conv2d(rev[-1], cfg.out_channels, 3, 1, 1)

    def nhwc(self, z):
        h = conv_nhwc(self.conv_in, z)
        h = self.mid_block.nhwc(h)
        for b in self.up_blocks:
            h = b.nhwc(h)
        return conv_nhwc(self.conv_out, h, norm=(self.conv_norm_out, True))     # (inference: the norm is applied inside the convolution where the library can)

    @ops.device_scoped
    def forward(self, z):
        return to_nchw_view(self.nhwc(to_nhwc(z)))


class AutoencoderKL(nn.Module):
    config_name = "config.json"
    weights_name = "diffusion_pytorch_model.safetensors"

    def __init__(self, **kwargs):
        super().__init__()
        cfg = Config(SD_VAE_CONFIG)
        cfg.update(kwargs)
        cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
        self.config = cfg
        self.encoder = Encoder(cfg)
        self.decoder = Decoder(cfg)
        self.quant_conv = Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
        self.post_quant_conv = Conv2d(cfg.latent_channels, cfg.latent_channels, 1)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def enable_gradient_checkpointing(self):
        """recompute the decoder's ResNet blocks in the backward (diffusers' AutoencoderKL supports the switch; the reference's training
        never flips it, train.py:304 only freezes the VAE): the frozen decoder still has to keep every activation for its input
        gradient — 3-5 GB per 576x576 image in fp32 — and this drops them"""
        for m in self.decoder.up_blocks:
            m.gradient_checkpointing = True

    def disable_gradient_checkpointing(self):
        for m in self.decoder.up_blocks:
            m.gradient_checkpointing = False

    def save_pretrained(self, save_directory, **kw):
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        cfg = dict(self.config)
        cfg["_class_name"] = "AutoencoderKL"
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}, os.path.join(save_directory, self.weights_name))

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, variant=None, **kw):
        from safetensors.torch import load_file
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, cls.config_name)) as f:
            cfg = {k: v for k, v in json.load(f).items() if k in SD_VAE_CONFIG}
        m = cls(**cfg)
        from .unet import load_weights
        sd = load_weights(d, cls.weights_name, variant)
        # older checkpoints name the mid-block attention projections query/key/value/proj_attn
        ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
        for k in list(sd):
            for old, new in ren.items():
                if ".attentions.0." + old + "." in k:
                    v = sd.pop(k)
                    sd[k.replace(".attentions.0." + old + ".", ".attentions.0." + new + ".")] = v.squeeze(-1).squeeze(-1) if v.dim() == 4 else v
        m.load_state_dict(sd)
        return m.to(torch_dtype) if torch_dtype is not None else m
