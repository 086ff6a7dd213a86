"""Pure-torch (CPU, NCHW) restatement of the SD-v2 / GeoWizard `UNet2DConditionModel` forward.  TEST INFRASTRUCTURE ONLY.

Functional over a diffusers-layout state dict `sd` (key names identical to diffusers so that real checkpoints load).
Reference wiring (paths relative to /root/reference/GeoWizard/geowizard/models):
  forward                      unet_2d_condition.py:916-1221
  CrossAttnDownBlock2D         unet_2d_blocks.py:1027-1185      DownBlock2D   :1188-1273
  UNetMidBlock2DCrossAttn      unet_2d_blocks.py:634-777
  CrossAttnUpBlock2D           unet_2d_blocks.py:2201-2371      UpBlock2D     :2374-2481
  Transformer2DModel           transformer_2d.py:326-347,407-423
  BasicTransformerBlock        attention.py:292-413             FeedForward   :719-777
  XFormersJointAttnProcessor   attention.py:425-513
Leaf modules follow diffusers==0.30.2 (not in tree): ResnetBlock2D, Attention/AttnProcessor2_0, GEGLU, Timesteps,
TimestepEmbedding, Downsample2D (conv s2 p1), Upsample2D (nearest + conv).
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------------------------
# state-dict specification (key -> shape), independent of the product's nn.Module tree
def unet_param_shapes(cfg):
    boc = cfg["block_out_channels"]
    temb = boc[0] * 4
    X = cfg["cross_attention_dim"]
    shapes = {}

    def conv(name, co, ci, k):
        shapes[name + ".weight"] = (co, ci, k, k)
        shapes[name + ".bias"] = (co,)

    def lin(name, co, ci, bias=True):
        shapes[name + ".weight"] = (co, ci)
        if bias:
            shapes[name + ".bias"] = (co,)

    def norm(name, c):
        shapes[name + ".weight"] = (c,)
        shapes[name + ".bias"] = (c,)

    def resnet(name, ci, co):
        norm(name + ".norm1", ci)
        conv(name + ".conv1", co, ci, 3)
        lin(name + ".time_emb_proj", co, temb)
        norm(name + ".norm2", co)
        conv(name + ".conv2", co, co, 3)
        if ci != co:
            conv(name + ".conv_shortcut", co, ci, 1)

    def transformer(name, c):
        norm(name + ".norm", c)
        lin(name + ".proj_in", c, c)
        b = name + ".transformer_blocks.0"
        norm(b + ".norm1", c)
        for q in ("to_q", "to_k", "to_v"):
            lin(b + ".attn1." + q, c, c, bias=False)
        lin(b + ".attn1.to_out.0", c, c)
        norm(b + ".norm2", c)
        lin(b + ".attn2.to_q", c, c, bias=False)
        lin(b + ".attn2.to_k", c, X, bias=False)
        lin(b + ".attn2.to_v", c, X, bias=False)
        lin(b + ".attn2.to_out.0", c, c)
        norm(b + ".norm3", c)
        lin(b + ".ff.net.0.proj", 8 * c, c)
        lin(b + ".ff.net.2", c, 4 * c)
        lin(name + ".proj_out", c, c)

    conv("conv_in", boc[0], cfg["in_channels"], 3)
    lin("time_embedding.linear_1", temb, boc[0])
    lin("time_embedding.linear_2", temb, temb)
    if cfg.get("class_embed_type") == "projection":
        lin("class_embedding.linear_1", temb, cfg["projection_class_embeddings_input_dim"])
        lin("class_embedding.linear_2", temb, temb)
    L = cfg["layers_per_block"]
    # down
    out_c = boc[0]
    for i, bt in enumerate(cfg["down_block_types"]):
        in_c, out_c = out_c, boc[i]
        for j in range(L):
            resnet("down_blocks.%d.resnets.%d" % (i, j), in_c if j == 0 else out_c, out_c)
            if bt == "CrossAttnDownBlock2D":
                transformer("down_blocks.%d.attentions.%d" % (i, j), out_c)
        if i != len(boc) - 1:
            conv("down_blocks.%d.downsamplers.0.conv" % i, out_c, out_c, 3)
    # mid
    resnet("mid_block.resnets.0", boc[-1], boc[-1])
    transformer("mid_block.attentions.0", boc[-1])
    resnet("mid_block.resnets.1", boc[-1], boc[-1])
    # up
    rev = list(reversed(boc))
    out_c = rev[0]
    for i, bt in enumerate(cfg["up_block_types"]):
        prev_out = out_c
        out_c = rev[i]
        in_c = rev[min(i + 1, len(boc) - 1)]
        for j in range(L + 1):
            skip_c = in_c if j == L else out_c
            res_in = prev_out if j == 0 else out_c
            resnet("up_blocks.%d.resnets.%d" % (i, j), res_in + skip_c, out_c)
            if bt == "CrossAttnUpBlock2D":
                transformer("up_blocks.%d.attentions.%d" % (i, j), out_c)
        if i != len(boc) - 1:
            conv("up_blocks.%d.upsamplers.0.conv" % i, out_c, out_c, 3)
    norm("conv_norm_out", boc[0])
    conv("conv_out", cfg["out_channels"], boc[0], 3)
    return shapes


# ----------------------------------------------------------------------------------------------------------------
def timestep_sinusoid(timesteps, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000):
    """diffusers get_timestep_embedding (embeddings.py); always fp32."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _conv(sd, name, x, stride=1, padding=1):
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=padding)


def _gn(sd, name, x, groups, eps):
    return F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps)


def _ln(sd, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def resnet_block(sd, name, x, temb, groups, eps):
    """diffusers ResnetBlock2D (time_embedding_norm='default', output_scale_factor=1, dropout 0)."""
    h = F.silu(_gn(sd, name + ".norm1", x, groups, eps))
    h = _conv(sd, name + ".conv1", h)
    if temb is not None and (name + ".time_emb_proj.weight") in sd:
        h = h + _lin(sd, name + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.silu(_gn(sd, name + ".norm2", h, groups, eps))
    h = _conv(sd, name + ".conv2", h)
    if (name + ".conv_shortcut.weight") in sd:
        x = _conv(sd, name + ".conv_shortcut", x, padding=0)
    return x + h


def attention(sd, name, x, ctx, heads, joint=False):
    """diffusers Attention with AttnProcessor2_0: q,k,v no bias; scale = head_dim^-0.5; to_out.0 with bias.
    joint=True: GeoWizard cross-domain self-attention (attention.py:482-491): keys/values of the two task halves of the
    batch are concatenated along the sequence axis and shared by both halves."""
    B, N, Cc = x.shape
    src = x if ctx is None else ctx
    q = _lin(sd, name + ".to_q", x)
    k = _lin(sd, name + ".to_k", src)
    v = _lin(sd, name + ".to_v", src)
    if joint:
        k0, k1 = torch.chunk(k, 2, dim=0)
        v0, v1 = torch.chunk(v, 2, dim=0)
        k = torch.cat([torch.cat([k0, k1], dim=1)] * 2, dim=0)
        v = torch.cat([torch.cat([v0, v1], dim=1)] * 2, dim=0)
    d = Cc // heads

    def split(t):
        return t.reshape(t.shape[0], t.shape[1], heads, d).transpose(1, 2)

    o = F.scaled_dot_product_attention(split(q), split(k), split(v))
    o = o.transpose(1, 2).reshape(B, N, Cc)
    return _lin(sd, name + ".to_out.0", o)


def transformer_block(sd, name, h, ctx, heads, joint):
    """BasicTransformerBlock (attention.py:292-413): LN -> self-attn -> + ; LN -> cross-attn -> + ; LN -> GEGLU FF -> +"""
    h = h + attention(sd, name + ".attn1", _ln(sd, name + ".norm1", h), None, heads, joint=joint)
    h = h + attention(sd, name + ".attn2", _ln(sd, name + ".norm2", h), ctx, heads)
    n = _ln(sd, name + ".norm3", h)
    g = _lin(sd, name + ".ff.net.0.proj", n)
    a, gate = g.chunk(2, dim=-1)
    return h + _lin(sd, name + ".ff.net.2", a * F.gelu(gate))


def transformer_2d(sd, name, x, ctx, heads, groups, joint):
    """Transformer2DModel, continuous input, use_linear_projection=True (transformer_2d.py:326-347,407-423)."""
    B, Cc, H, W = x.shape
    res = x
    h = _gn(sd, name + ".norm", x, groups, 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, Cc)
    h = _lin(sd, name + ".proj_in", h)
    h = transformer_block(sd, name + ".transformer_blocks.0", h, ctx, heads, joint)
    h = _lin(sd, name + ".proj_out", h)
    h = h.reshape(B, H, W, Cc).permute(0, 3, 1, 2)
    return h + res


def unet_forward(sd, cfg, sample, timestep, encoder_hidden_states, class_labels=None):
    """UNet2DConditionModel.forward (unet_2d_condition.py:916-1221). sample [B,Cin,h,w]; timestep scalar or [B];
    encoder_hidden_states [B,L,X]; class_labels [B,P] for the GeoWizard projection class embedding."""
    boc = cfg["block_out_channels"]
    groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    heads = cfg["attention_head_dim"]
    joint = bool(cfg.get("joint_attention"))
    L = cfg["layers_per_block"]
    n_up = len(boc) - 1
    forward_upsample_size = any(s % (2 ** n_up) != 0 for s in sample.shape[-2:])

    # 1. time (+ class) embedding (:960-1000)
    t = torch.as_tensor(timestep, device=sample.device)
    if t.dim() == 0:
        t = t[None]
    t = t.expand(sample.shape[0])
    t_emb = timestep_sinusoid(t, boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"]).to(sample.dtype)
    emb = _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", t_emb)))
    if cfg.get("class_embed_type") == "projection":
        cl = class_labels.to(sample.dtype)
        emb = emb + _lin(sd, "class_embedding.linear_2", F.silu(_lin(sd, "class_embedding.linear_1", cl)))

    # 2. conv_in (:1084)
    h = _conv(sd, "conv_in", sample)
    skips = [h]
    # 3. down (:1117-1138)
    for i, bt in enumerate(cfg["down_block_types"]):
        for j in range(L):
            h = resnet_block(sd, "down_blocks.%d.resnets.%d" % (i, j), h, emb, groups, eps)
            if bt == "CrossAttnDownBlock2D":
                h = transformer_2d(sd, "down_blocks.%d.attentions.%d" % (i, j), h, encoder_hidden_states, heads[i], groups, joint)
            skips.append(h)
        if i != len(boc) - 1:
            h = _conv(sd, "down_blocks.%d.downsamplers.0.conv" % i, h, stride=2, padding=1)
            skips.append(h)
    # 4. mid (:1152-1161)
    h = resnet_block(sd, "mid_block.resnets.0", h, emb, groups, eps)
    h = transformer_2d(sd, "mid_block.attentions.0", h, encoder_hidden_states, heads[-1], groups, joint)
    h = resnet_block(sd, "mid_block.resnets.1", h, emb, groups, eps)
    # 5. up (:1177-1206)
    rheads = list(reversed(heads))
    for i, bt in enumerate(cfg["up_block_types"]):
        res = skips[-(L + 1):]
        skips = skips[:-(L + 1)]
        for j in range(L + 1):
            h = torch.cat([h, res.pop()], dim=1)
            h = resnet_block(sd, "up_blocks.%d.resnets.%d" % (i, j), h, emb, groups, eps)
            if bt == "CrossAttnUpBlock2D":
                h = transformer_2d(sd, "up_blocks.%d.attentions.%d" % (i, j), h, encoder_hidden_states, rheads[i], groups, joint)
        if i != len(boc) - 1:
            if forward_upsample_size:
                h = F.interpolate(h, size=skips[-1].shape[2:], mode="nearest")
            else:
                h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, "up_blocks.%d.upsamplers.0.conv" % i, h)
    # 6. out (:1209-1212)
    h = F.silu(_gn(sd, "conv_norm_out", h, groups, eps))
    return _conv(sd, "conv_out", h)
