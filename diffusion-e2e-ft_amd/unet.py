"""SD-v2 / GeoWizard `UNet2DConditionModel` on libe2eft: same constructor config, parameter names, call signature
and attribute surface the reference's callers use:
  unet(sample, timestep, encoder_hidden_states[, class_labels=...][, return_dict=False]) -> .sample / [0]
      Marigold/marigold/marigold_pipeline.py:452-454; training/train.py:500; geowizard_pipeline.py:319-321
  .conv_in (nn.Conv2d), .config['in_channels']      training/util/unet_prep.py:7-20, training/train.py:299
  .enable_gradient_checkpointing(), .enable_xformers_memory_efficient_attention(), .save_pretrained / from_pretrained
Wiring follows GeoWizard/geowizard/models/unet_2d_condition.py:916-1221 and unet_2d_blocks.py (block classes cited below).
"""
import json
import os
from collections import OrderedDict

import torch
from torch import nn

from . import ops
from . import autograd as F
from .modules import (Conv2d, GroupNorm, ResnetBlock2D, Transformer2DModel, Downsample2D, Upsample2D, TimestepEmbedding, TimeCond, CtxCond,
                      checkpointed, conv_nhwc, to_nhwc, to_nchw_view)


BATCHED_PROJECTIONS = True   # A/B switch of UNet2DConditionModel._batch_small_gemms (inference: time / context projections as two GEMMs)


class Config(OrderedDict):
    """dict with attribute access (diffusers FrozenDict surface: cfg.x, cfg['x'], cfg['x'] = v)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


class UNet2DConditionOutput:
    def __init__(self, sample):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]


SD2_UNET_CONFIG = dict(
    sample_size=96, in_channels=4, out_channels=4,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, attention_head_dim=(5, 10, 20, 20),
    cross_attention_dim=1024, norm_num_groups=32, norm_eps=1e-5, use_linear_projection=True, flip_sin_to_cos=True,
    freq_shift=0, class_embed_type=None, projection_class_embeddings_input_dim=None, joint_attention=False,
)


class _DownBlock(nn.Module):
    """CrossAttnDownBlock2D (unet_2d_blocks.py:1027-1185) / DownBlock2D (:1188-1273)."""

    def __init__(self, in_c, out_c, temb, layers, groups, eps, heads, xdim, cross, add_down, joint):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_c if j == 0 else out_c, out_c, temb, groups, eps) for j in range(layers)])
        if cross:
            self.attentions = nn.ModuleList([Transformer2DModel(out_c, heads, xdim, groups, joint) for _ in range(layers)])
        self.has_cross_attention = cross
        if add_down:
            self.downsamplers = nn.ModuleList([Downsample2D(out_c, padding=1)])
        self.add_down = add_down
        self.gradient_checkpointing = False

    def nhwc(self, h, temb_act, ctx):
        outs = []
        ck = self.gradient_checkpointing and self.training
        for j, res in enumerate(self.resnets):
            h = checkpointed(ck, res.nhwc, h, temb_act)
            if self.has_cross_attention:
                h = checkpointed(ck, self.attentions[j].nhwc, h, ctx)
            outs.append(h)
        if self.add_down:
            h = self.downsamplers[0].nhwc(h)
            outs.append(h)
        return h, outs


class _MidBlock(nn.Module):
    """UNetMidBlock2DCrossAttn (unet_2d_blocks.py:634-777): resnet -> transformer -> resnet."""

    def __init__(self, c, temb, groups, eps, heads, xdim, joint):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, groups, eps), ResnetBlock2D(c, c, temb, groups, eps)])
        self.attentions = nn.ModuleList([Transformer2DModel(c, heads, xdim, groups, joint)])
        self.gradient_checkpointing = False

    def nhwc(self, h, temb_act, ctx):
        ck = self.gradient_checkpointing and self.training
        h = checkpointed(ck, self.resnets[0].nhwc, h, temb_act)
        h = checkpointed(ck, self.attentions[0].nhwc, h, ctx)
        return checkpointed(ck, self.resnets[1].nhwc, h, temb_act)


class _UpBlock(nn.Module):
    """CrossAttnUpBlock2D (unet_2d_blocks.py:2201-2371) / UpBlock2D (:2374-2481).  torch.cat([hidden, skip], dim=1)
    (:2328,2456) is never materialised: GroupNorm, conv1 and the 1x1 shortcut read the two sources directly."""

    def __init__(self, in_c, out_c, prev_c, temb, layers, groups, eps, heads, xdim, cross, add_up, joint):
        super().__init__()
        rs = []
        for j in range(layers):
            skip_c = in_c if j == layers - 1 else out_c
            res_in = prev_c if j == 0 else out_c
            rs.append(ResnetBlock2D(res_in + skip_c, out_c, temb, groups, eps))
        self.resnets = nn.ModuleList(rs)
        if cross:
            self.attentions = nn.ModuleList([Transformer2DModel(out_c, heads, xdim, groups, joint) for _ in range(layers)])
        self.has_cross_attention = cross
        if add_up:
            self.upsamplers = nn.ModuleList([Upsample2D(out_c)])
        self.add_up = add_up
        self.gradient_checkpointing = False

    def nhwc(self, h, skips, temb_act, ctx, upsample_size=None):
        ck = self.gradient_checkpointing and self.training
        for j, res in enumerate(self.resnets):
            h = checkpointed(ck, res.nhwc, h, temb_act, skips.pop())
            if self.has_cross_attention:
                h = checkpointed(ck, self.attentions[j].nhwc, h, ctx)
        if self.add_up:
            h = self.upsamplers[0].nhwc(h, size=upsample_size)
        return h


def _weights_file(d, name, variant=None):
    """diffusers naming: diffusion_pytorch_model[.<variant>].safetensors (run.py:256-282 passes variant="fp16" with --half_precision)"""
    if variant:
        stem, ext = os.path.splitext(name)
        cand = os.path.join(d, "%s.%s%s" % (stem, variant, ext))
        if os.path.exists(cand):
            return cand
    return os.path.join(d, name)


# diffusers UNet2DConditionModel config keys this implementation does not carry, with the only value it implements.  A checkpoint
# whose config.json says otherwise would load into the wrong architecture: refuse it (unknown keys with a None / False / default
# value are ignored, as diffusers ignores keys it does not know).
IMPLEMENTED_ONLY = dict(center_input_sample=False, mid_block_type="UNetMidBlock2DCrossAttn", only_cross_attention=False, downsample_padding=1,
                        mid_block_scale_factor=1, dropout=0.0, act_fn="silu", transformer_layers_per_block=1, encoder_hid_dim=None,
                        encoder_hid_dim_type=None, num_attention_heads=None, dual_cross_attention=False, addition_embed_type=None,
                        addition_time_embed_dim=None, num_class_embeds=None, resnet_time_scale_shift="default",
                        resnet_skip_time_act=False, resnet_out_scale_factor=1.0, time_embedding_type="positional", time_embedding_dim=None,
                        time_embedding_act_fn=None, timestep_post_act=None, time_cond_proj_dim=None, conv_in_kernel=3, conv_out_kernel=3,
                        attention_type="default", class_embeddings_concat=False, mid_block_only_cross_attention=None,
                        cross_attention_norm=None, reverse_transformer_layers_per_block=None)


def check_unsupported_config(raw):
    # (`upcast_attention` is accepted either way: softmax statistics and accumulation are fp32 in every attention kernel here)
    bad = {k: v for k, v in raw.items() if k in IMPLEMENTED_ONLY and v != IMPLEMENTED_ONLY[k]}
    if bad:
        raise NotImplementedError("UNet config.json asks for features this implementation does not have (SD-v2 / Marigold / GeoWizard "
                                  "configurations only): %s" % bad)


def load_weights(d, weights_name, variant=None):
    """diffusion_pytorch_model[.<variant>].safetensors, else the pickled .bin twin diffusers falls back to"""
    f = _weights_file(d, weights_name, variant)
    if os.path.exists(f):
        from safetensors.torch import load_file
        return load_file(f)
    b = _weights_file(d, weights_name.replace(".safetensors", ".bin"), variant)
    if os.path.exists(b):
        return torch.load(b, map_location="cpu", weights_only=True)
    raise FileNotFoundError("no %s (or .bin) under %s" % (weights_name, d))


class UNet2DConditionModel(nn.Module):
    config_name = "config.json"
    weights_name = "diffusion_pytorch_model.safetensors"

    def __init__(self, **kwargs):
        super().__init__()
        cfg = Config(SD2_UNET_CONFIG)
        cfg.update(kwargs)
        for k in ("block_out_channels", "attention_head_dim", "down_block_types", "up_block_types"):
            cfg[k] = tuple(cfg[k]) if isinstance(cfg[k], (list, tuple)) else (cfg[k],) * len(cfg["block_out_channels"])
        if not cfg["use_linear_projection"]:
            raise NotImplementedError("only use_linear_projection=True (SD-v2) is implemented")
        self.config = cfg
        boc, heads = cfg.block_out_channels, cfg.attention_head_dim
        g, eps, X, L = cfg.norm_num_groups, cfg.norm_eps, cfg.cross_attention_dim, cfg.layers_per_block
        joint = bool(cfg.joint_attention)
        temb = boc[0] * 4
        self.conv_in = Conv2d(cfg.in_channels, boc[0], 3, 1, 1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        if cfg.class_embed_type == "projection":
            self.class_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, temb)
        elif cfg.class_embed_type is None:
            self.class_embedding = None
        else:
            raise NotImplementedError("class_embed_type=%r" % (cfg.class_embed_type,))
        downs = []
        out_c = boc[0]
        for i, bt in enumerate(cfg.down_block_types):
            in_c, out_c = out_c, boc[i]
            downs.append(_DownBlock(in_c, out_c, temb, L, g, eps, heads[i], X, bt == "CrossAttnDownBlock2D", i != len(boc) - 1, joint))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = _MidBlock(boc[-1], temb, g, eps, heads[-1], X, joint)
        ups = []
        rev, rheads = list(reversed(boc)), list(reversed(heads))
        out_c = rev[0]
        for i, bt in enumerate(cfg.up_block_types):
            prev, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            ups.append(_UpBlock(in_c, out_c, prev, temb, L + 1, g, eps, rheads[i], X, bt == "CrossAttnUpBlock2D", i != len(boc) - 1, joint))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = GroupNorm(g, boc[0], eps=eps, affine=True)
        self.conv_out = Conv2d(boc[0], cfg.out_channels, 3, 1, 1)
        self.gradient_checkpointing = False

    # ---- attribute surface used by the reference's callers ----
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def compute_dtype(self):
        """dtype of the activations: the parameters' dtype unless set_compute_dtype() asked for 16-bit compute over fp32 master
        weights (the kernels cast each parameter once per version)"""
        return getattr(self, "_compute_dtype", None) or self.dtype

    def set_compute_dtype(self, dtype):
        self._compute_dtype = dtype
        return self

    def enable_gradient_checkpointing(self):  # training/train.py:342-343 (--gradient_checkpointing in every training script)
        """per-block activation recompute in training mode, as diffusers' `_set_gradient_checkpointing` flips it on every block
        (unet_2d_condition.py `_supports_gradient_checkpointing`, unet_2d_blocks.py:1136-1161); see modules.checkpointed"""
        self._set_gradient_checkpointing(True)

    def disable_gradient_checkpointing(self):
        self._set_gradient_checkpointing(False)

    def _set_gradient_checkpointing(self, value):
        self.gradient_checkpointing = value
        for m in self.modules():
            if isinstance(m, (_DownBlock, _MidBlock, _UpBlock)):
                m.gradient_checkpointing = value

    def enable_xformers_memory_efficient_attention(self, *a, **k):  # train.py:317, run.py:284-289: fused attention is always on
        return None

    def register_to_config(self, **kw):  # training/train.py:334-336
        self.config.update(kw)

    def save_pretrained(self, save_directory, **kw):  # training/train.py:326,612-630
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        cfg = dict(self.config)
        cfg["_class_name"] = "UNet2DConditionModel"
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}, os.path.join(save_directory, self.weights_name))

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, variant=None, **kw):  # training/train.py:292-296
        from safetensors.torch import load_file
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, cls.config_name)) as f:
            raw = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        check_unsupported_config(raw)
        m = cls(**{k: v for k, v in raw.items() if k in SD2_UNET_CONFIG})
        m.load_state_dict(load_weights(d, cls.weights_name, variant))
        return m.to(torch_dtype) if torch_dtype is not None else m

    def _batch_small_gemms(self, temb_act, ctx, shared_src=None):
        """Inference only (no autograd graph).  The 22 `time_emb_proj` Linears (unet_2d_blocks.py: every ResnetBlock2D) all read the same
        [B, 1280] embedding and the 16 cross-attention `to_k` / `to_v` pairs all read the same [B, L, 1024] context: as separate launches
        they are ~40 GEMMs with 8-16 rows, each a 25-30 us latency-bound k-loop on a handful of workgroups (~1 ms of a 112 ms step).
        Here each family is ONE GEMM over the weights concatenated along the output dimension (the same dot products, element for
        element); the consumers pick up their slice (`ResnetBlock2D.nhwc`, `Attention.forward`)."""
        from .modules import ResnetBlock2D, Attention
        from . import modules as _M
        # a two-token context shared by the whole batch: the cross-attention layers fold it (modules.Attention._fold) — independent of the batching switch below
        fold = _M.CROSS_ATTN_FOLD and shared_src is not None and shared_src.shape[1] == 2 and not self.config.joint_attention
        if not BATCHED_PROJECTIONS:    # A/B switch
            return temb_act, (CtxCond(ctx, None, shared=True, src=shared_src) if fold else ctx)
        dt = temb_act.dtype
        fam = self.__dict__.get("_small_gemm_family")
        if fam is None:
            res = [m for m in self.modules() if isinstance(m, ResnetBlock2D) and m.time_emb_proj is not None]
            att = [m for m in self.modules() if isinstance(m, Attention) and m.to_k.in_features != m.to_q.in_features
                   and m.to_k.bias is None and m.to_q.in_features // m.heads == 64]
            fam = self.__dict__["_small_gemm_family"] = (res, att)       # (which modules take part: structure, not per-call state)
        res, att = fam
        tc, cc = temb_act, ctx
        if res:
            ws = tuple(m.time_emb_proj.weight for m in res)
            bs = tuple(m.time_emb_proj.bias for m in res)
            bias = F.cached(self, "b_time_proj_all_%s" % dt, bs, lambda: torch.cat([b.detach().to(dt) for b in bs]))
            out = F.linear(temb_act, ws, bias, owner=self, name="w_time_proj_all")
            rows, o = {}, 0
            for m in res:
                c = m.time_emb_proj.out_features
                rows[id(m)] = out[:, o:o + c].contiguous()
                o += c
            tc = TimeCond(temb_act, rows)
        if att and dt != torch.float32 and not fold:
            ws = tuple(w for m in att for w in (m.to_k.weight, m.to_v.weight))
            kv = F.linear(ctx, ws, owner=self, name="w_ctx_kv_all")      # [B, L, sum 2C]
            kvs, o = {}, 0
            for m in att:
                c2 = 2 * m.to_k.out_features
                kvs[id(m)] = kv[..., o:o + c2]
                o += c2
            cc = CtxCond(ctx, kvs)
        elif fold:
            cc = CtxCond(ctx, None, shared=True, src=shared_src)       # (no k | v GEMM: the folding layers project the two tokens themselves, once per context)
        return tc, cc

    # ---- forward ----
    @ops.device_scoped
    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, return_dict=True, **unused):
        return self._forward(sample, timestep, encoder_hidden_states, class_labels, return_dict)

    def _forward(self, sample, timestep, encoder_hidden_states, class_labels, return_dict):
        cfg = self.config
        dt = self.compute_dtype
        if sample.dtype != dt:
            raise TypeError("sample dtype %s != model compute dtype %s" % (sample.dtype, dt))
        B = sample.shape[0]
        x = to_nhwc(sample)
        # 1. time (+ class) embedding  (unet_2d_condition.py:960-1000)
        t = torch.as_tensor(timestep, device=sample.device)
        if t.dim() == 0:
            t = t[None]
        t = t.expand(B).to(torch.int64).contiguous()
        emb = self.time_embedding(ops.timestep_embedding(t, cfg.block_out_channels[0], dt))
        if self.class_embedding is not None:
            if class_labels is None:
                raise ValueError("class_labels should be provided for class_embed_type='projection'")
            emb = F.add(emb, self.class_embedding(class_labels.to(dt).contiguous()))
        temb_act = F.silu(emb)  # every ResnetBlock2D applies SiLU before its time_emb_proj
        # one context for the whole batch (the pipelines pass the empty-prompt embedding as a stride-0 expand): cross-attention layers may fold it (modules.Attention._fold)
        ehs = encoder_hidden_states
        shared_src = ehs[:1] if (ehs.dim() == 3 and ehs.dtype == dt and (B == 1 or ehs.stride(0) == 0)) else None
        ctx = ehs.to(dt).contiguous()
        if not torch.is_grad_enabled():
            temb_act, ctx = self._batch_small_gemms(temb_act, ctx, shared_src)
        n_up = len(cfg.block_out_channels) - 1
        forward_upsample_size = any(s % (2 ** n_up) != 0 for s in sample.shape[-2:])
        # 2-3. conv_in, down
        h = conv_nhwc(self.conv_in, x)
        skips = [h]
        for blk in self.down_blocks:
            h, outs = blk.nhwc(h, temb_act, ctx)
            skips.extend(outs)
        # 4. mid
        h = self.mid_block.nhwc(h, temb_act, ctx)
        # 5. up
        for blk in self.up_blocks:
            n = len(blk.resnets)
            res = skips[-n:]
            skips = skips[:-n]
            size = skips[-1].shape[1:3] if (forward_upsample_size and blk.add_up) else None
            h = blk.nhwc(h, res, temb_act, ctx, upsample_size=size)
        # 6. out
        h = self.conv_norm_out.nhwc(h, silu=True)
        out = to_nchw_view(conv_nhwc(self.conv_out, h))
        return UNet2DConditionOutput(out) if return_dict else (out,)
