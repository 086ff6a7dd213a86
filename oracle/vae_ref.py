"""Pure-torch (CPU, NCHW) restatement of the SD `AutoencoderKL` pieces the pipelines call.  TEST INFRASTRUCTURE ONLY.

The reference calls four sub-modules directly and takes the mean half of the moments:
  Marigold/marigold/marigold_pipeline.py:493-497 (encoder, quant_conv), :513-519 (post_quant_conv, decoder);
  training/train.py:233-243.
Module semantics follow diffusers==0.30.2 (autoencoders/vae.py Encoder/Decoder, unet_2d_blocks.py
DownEncoderBlock2D/UpDecoderBlock2D/UNetMidBlock2D — vendored twins at
GeoWizard/geowizard/models/unet_2d_blocks.py:509-631,1276-1333,2484-2541): ResNet blocks without time embedding,
GroupNorm eps 1e-6, VAE downsample = F.pad(x,(0,1,0,1)) + conv s2 p0, mid-block single-head attention with bias,
group norm and residual inside the attention module."""
import torch
import torch.nn.functional as F

from .unet_ref import _conv, _gn, _lin, resnet_block


def vae_param_shapes(cfg):
    boc = cfg["block_out_channels"]
    L = cfg["layers_per_block"]
    lc = cfg["latent_channels"]
    shapes = {}

    def conv(name, co, ci, k):
        shapes[name + ".weight"] = (co, ci, k, k)
        shapes[name + ".bias"] = (co,)

    def norm(name, c):
        shapes[name + ".weight"] = (c,)
        shapes[name + ".bias"] = (c,)

    def resnet(name, ci, co):
        norm(name + ".norm1", ci)
        conv(name + ".conv1", co, ci, 3)
        norm(name + ".norm2", co)
        conv(name + ".conv2", co, co, 3)
        if ci != co:
            conv(name + ".conv_shortcut", co, ci, 1)

    def mid(name, c):
        resnet(name + ".resnets.0", c, c)
        norm(name + ".attentions.0.group_norm", c)
        for q in ("to_q", "to_k", "to_v", "to_out.0"):
            shapes[name + ".attentions.0." + q + ".weight"] = (c, c)
            shapes[name + ".attentions.0." + q + ".bias"] = (c,)
        resnet(name + ".resnets.1", c, c)

    # encoder
    conv("encoder.conv_in", boc[0], cfg["in_channels"], 3)
    out_c = boc[0]
    for i in range(len(boc)):
        in_c, out_c = out_c, boc[i]
        for j in range(L):
            resnet("encoder.down_blocks.%d.resnets.%d" % (i, j), in_c if j == 0 else out_c, out_c)
        if i != len(boc) - 1:
            conv("encoder.down_blocks.%d.downsamplers.0.conv" % i, out_c, out_c, 3)
    mid("encoder.mid_block", boc[-1])
    norm("encoder.conv_norm_out", boc[-1])
    conv("encoder.conv_out", 2 * lc, boc[-1], 3)
    conv("quant_conv", 2 * lc, 2 * lc, 1)
    conv("post_quant_conv", lc, lc, 1)
    # decoder
    rev = list(reversed(boc))
    conv("decoder.conv_in", rev[0], lc, 3)
    mid("decoder.mid_block", rev[0])
    out_c = rev[0]
    for i in range(len(boc)):
        prev, out_c = out_c, rev[i]
        for j in range(L + 1):
            resnet("decoder.up_blocks.%d.resnets.%d" % (i, j), prev if j == 0 else out_c, out_c)
        if i != len(boc) - 1:
            conv("decoder.up_blocks.%d.upsamplers.0.conv" % i, out_c, out_c, 3)
    norm("decoder.conv_norm_out", rev[-1])
    conv("decoder.conv_out", cfg["out_channels"], rev[-1], 3)
    return shapes


def vae_attention(sd, name, x, groups):
    """diffusers Attention as used by UNetMidBlock2D in the VAE: 1 head of dim C, bias, GroupNorm(eps 1e-6) on the
    tokens, residual_connection=True, rescale_output_factor=1."""
    B, Cc, H, W = x.shape
    res = x
    h = _gn(sd, name + ".group_norm", x.reshape(B, Cc, H * W), groups, 1e-6)
    h = h.transpose(1, 2)  # [B, HW, C]
    q = _lin(sd, name + ".to_q", h)
    k = _lin(sd, name + ".to_k", h)
    v = _lin(sd, name + ".to_v", h)
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    o = _lin(sd, name + ".to_out.0", o)
    return o.transpose(1, 2).reshape(B, Cc, H, W) + res


def _mid(sd, name, h, groups):
    h = resnet_block(sd, name + ".resnets.0", h, None, groups, 1e-6)
    h = vae_attention(sd, name + ".attentions.0", h, groups)
    return resnet_block(sd, name + ".resnets.1", h, None, groups, 1e-6)


def encoder_forward(sd, cfg, x):
    """vae.encoder(x): [B,3,H,W] -> [B,2*latent,H/8,W/8]"""
    boc = cfg["block_out_channels"]
    g = cfg["norm_num_groups"]
    h = _conv(sd, "encoder.conv_in", x)
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"]):
            h = resnet_block(sd, "encoder.down_blocks.%d.resnets.%d" % (i, j), h, None, g, 1e-6)
        if i != len(boc) - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = _conv(sd, "encoder.down_blocks.%d.downsamplers.0.conv" % i, h, stride=2, padding=0)
    h = _mid(sd, "encoder.mid_block", h, g)
    h = F.silu(_gn(sd, "encoder.conv_norm_out", h, g, 1e-6))
    return _conv(sd, "encoder.conv_out", h)


def quant_conv(sd, h):
    return _conv(sd, "quant_conv", h, padding=0)


def post_quant_conv(sd, z):
    return _conv(sd, "post_quant_conv", z, padding=0)


def decoder_forward(sd, cfg, z):
    """vae.decoder(z): [B,latent,h,w] -> [B,3,8h,8w]"""
    boc = cfg["block_out_channels"]
    g = cfg["norm_num_groups"]
    h = _conv(sd, "decoder.conv_in", z)
    h = _mid(sd, "decoder.mid_block", h, g)
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"] + 1):
            h = resnet_block(sd, "decoder.up_blocks.%d.resnets.%d" % (i, j), h, None, g, 1e-6)
        if i != len(boc) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, "decoder.up_blocks.%d.upsamplers.0.conv" % i, h)
    h = F.silu(_gn(sd, "decoder.conv_norm_out", h, g, 1e-6))
    return _conv(sd, "decoder.conv_out", h)
