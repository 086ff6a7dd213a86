"""CLIP ViT image encoder on libe2eft — GeoWizard's per-image conditioning.

Replaces `transformers.CLIPVisionModelWithProjection` as used by
/root/reference/GeoWizard/geowizard/models/geowizard_pipeline.py:76-86,232-248 (`self.image_encoder(img).image_embeds`): same
constructor config keys, same state-dict layout (`vision_model.embeddings.*`, `vision_model.encoder.layers.N.*`,
`vision_model.pre_layrnorm` [sic], `vision_model.post_layernorm`, `visual_projection`), same output attribute.  Inference only (the
encoder is frozen everywhere in the reference).  ViT-L/14: 24 layers x (LN -> 16-head attention d=64 -> +, LN -> 1024-4096-1024
quick-GELU MLP -> +) over 257 tokens; every matmul runs on the implicit-GEMM kernel (residuals folded into the epilogue), the
attention core on the fused d=64 kernel, LayerNorm / activation as HBM streams.
"""
import json
import os

import torch
import torch.nn as nn

from . import autograd as F
from . import ops
from .modules import Conv2d, LayerNorm, Linear, conv_nhwc, to_nhwc

CLIP_VIT_L14 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=224,
                    patch_size=14, projection_dim=768, hidden_act="quick_gelu", layer_norm_eps=1e-5, num_channels=3)
# transformers CLIPImageProcessor defaults (feature_extractor.image_mean / image_std / crop_size of the hub preprocessor config)
CLIP_IMAGE_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_IMAGE_STD = (0.26862954, 0.26130258, 0.27577711)


class _Out:
    def __init__(self, image_embeds, last_hidden_state):
        self.image_embeds, self.last_hidden_state = image_embeds, last_hidden_state

    def __getitem__(self, i):
        return (self.image_embeds, self.last_hidden_state)[i]


class CLIPAttention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads, self.scale = heads, (dim // heads) ** -0.5
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = Linear(dim, dim), Linear(dim, dim), Linear(dim, dim), Linear(dim, dim)

    def forward(self, x, residual):
        bias = F.cached(self, "bqkv_%s" % x.dtype, (self.q_proj.bias, self.k_proj.bias, self.v_proj.bias),
                        lambda: torch.cat([self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]).detach().to(x.dtype))
        qkv = F.linear(x, (self.q_proj.weight, self.k_proj.weight, self.v_proj.weight), bias, owner=self, name="wqkv")
        return self.out_proj(F.attention(qkv, None, self.heads, self.scale), residual=residual)


class CLIPMLP(nn.Module):
    def __init__(self, dim, inner, act):
        super().__init__()
        if act not in ops.ACT_KINDS:
            raise ValueError("unsupported CLIP hidden_act %r" % (act,))
        self.act = act
        self.fc1, self.fc2 = Linear(dim, inner), Linear(inner, dim)

    def forward(self, x, residual):
        return self.fc2(ops.activation(self.fc1(x), self.act), residual=residual)


class CLIPEncoderLayer(nn.Module):
    def __init__(self, dim, inner, heads, act, eps):
        super().__init__()
        self.self_attn = CLIPAttention(dim, heads)
        self.layer_norm1 = LayerNorm(dim, eps=eps)
        self.mlp = CLIPMLP(dim, inner, act)
        self.layer_norm2 = LayerNorm(dim, eps=eps)

    def forward(self, x):
        x = self.self_attn(self.layer_norm1(x), residual=x)
        return self.mlp(self.layer_norm2(x), residual=x)


class CLIPVisionEmbeddings(nn.Module):
    def __init__(self, dim, image_size, patch, channels):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.randn(dim))
        self.patch_embedding = Conv2d(channels, dim, kernel_size=patch, stride=patch, bias=False)
        self.num_positions = (image_size // patch) ** 2 + 1
        self.position_embedding = nn.Embedding(self.num_positions, dim)

    def forward(self, pixel_values):
        B = pixel_values.shape[0]
        dt = self.patch_embedding.weight.dtype
        p = conv_nhwc(self.patch_embedding, to_nhwc(pixel_values.to(dt)), gn_stats=False)      # [B, g, g, C]: the 14x14/14 conv is one GEMM, K = 14*14*8
        C = p.shape[-1]
        pos = self.position_embedding.weight
        x = torch.empty((B, self.num_positions, C), dtype=dt, device=p.device)
        x[:, 0] = self.class_embedding + pos[0]
        torch.add(p.reshape(B, -1, C), pos[1:], out=x[:, 1:])
        return x


class _Encoder(nn.Module):
    def __init__(self, n, *a):
        super().__init__()
        self.layers = nn.ModuleList([CLIPEncoderLayer(*a) for _ in range(n)])


class CLIPVisionTransformer(nn.Module):
    def __init__(self, c):
        super().__init__()
        dim = c["hidden_size"]
        self.embeddings = CLIPVisionEmbeddings(dim, c["image_size"], c["patch_size"], c["num_channels"])
        self.pre_layrnorm = LayerNorm(dim, eps=c["layer_norm_eps"])   # the misspelling is the checkpoint key
        self.encoder = _Encoder(c["num_hidden_layers"], dim, c["intermediate_size"], c["num_attention_heads"], c["hidden_act"], c["layer_norm_eps"])
        self.post_layernorm = LayerNorm(dim, eps=c["layer_norm_eps"])

    def forward(self, pixel_values):
        x = self.pre_layrnorm(self.embeddings(pixel_values))
        for layer in self.encoder.layers:
            x = layer(x)
        return x, self.post_layernorm(x[:, 0].contiguous())


class _HubIO:
    """transformers-style directory IO: config.json + model.safetensors (the layout of a diffusers checkpoint's text_encoder/ and
    image_encoder/ sub-folders; Marigold/run.py:270, GeoWizard run_infer.py)."""
    weights_name = "model.safetensors"
    _class_name = None

    def save_pretrained(self, save_directory, **kw):
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(dict(self.config, architectures=[self._class_name]), f, indent=2)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}, os.path.join(save_directory, self.weights_name))

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, variant=None, **kw):
        from safetensors.torch import load_file
        from .unet import _weights_file
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, "config.json")) as f:
            raw = json.load(f)
        raw = dict(raw.get(cls._config_section, {}), **{k: v for k, v in raw.items() if not isinstance(v, dict)}) if cls._config_section else raw
        m = cls(**{k: raw[k] for k in cls._defaults if k in raw})
        sd = load_file(_weights_file(d, cls.weights_name, variant))
        pre = cls._prefix
        if not any(k.startswith(pre) for k in sd):          # transformers >= 5 saves the tower without its wrapper prefix
            sd = {(k if k.startswith("visual_projection") else pre + k): v for k, v in sd.items()}
        m.load_state_dict(sd)
        return m.to(torch_dtype) if torch_dtype is not None else m


class CLIPVisionModelWithProjection(_HubIO, nn.Module):
    """`image_encoder` slot of DepthNormalEstimationPipeline (geowizard_pipeline.py:76-86)."""
    _class_name, _prefix, _config_section, _defaults = "CLIPVisionModelWithProjection", "vision_model.", "vision_config", CLIP_VIT_L14

    def __init__(self, **kwargs):
        super().__init__()
        cfg = dict(CLIP_VIT_L14)
        cfg.update(kwargs)
        self.config = cfg
        self.vision_model = CLIPVisionTransformer(cfg)
        self.visual_projection = Linear(cfg["hidden_size"], cfg["projection_dim"], bias=False)

    @property
    def dtype(self):
        return self.visual_projection.weight.dtype

    @property
    def device(self):
        return self.visual_projection.weight.device

    def load_state_dict(self, sd, strict=True, **kw):
        sd = {k: v for k, v in sd.items() if not k.endswith("position_ids")}   # a persistent buffer in older transformers releases
        return super().load_state_dict(sd, strict=strict, **kw)

    @torch.no_grad()
    @ops.device_scoped
    def forward(self, pixel_values, **unused):
        last, pooled = self.vision_model(pixel_values.to(self.device))
        return _Out(self.visual_projection(pooled), last)


def preprocess_for_clip(rgb, size=224, mean=CLIP_IMAGE_MEAN, std=CLIP_IMAGE_STD):
    """geowizard_pipeline.py:236-245: rgb in [-1, 1] [B,3,H,W] -> bicubic antialiased resize of (rgb+1)/2 to size x size, then
    (x - mean) / std in float32 (torchvision TF.resize(antialias=True) == F.interpolate(mode="bicubic", antialias=True))."""
    as_t = lambda v: v if isinstance(v, torch.Tensor) else torch.tensor(v, device=rgb.device, dtype=torch.float32)[:, None, None]
    if rgb.is_cuda:      # the table-driven resampler of csrc/prepost.hip with aten's antialiased-bicubic tables; (x + 1) / 2 rides on its store
        from .pipeline import resize_device
        B, C3, H, W = rgb.shape
        with ops.on_device_of(rgb):
            x = resize_device(rgb.reshape(B * C3, H, W), (size, size), mul=0.5, add=0.5, kind="bicubic").view(B, C3, size, size)
    else:
        x = torch.nn.functional.interpolate((rgb + 1) / 2, size=(size, size), mode="bicubic", antialias=True, align_corners=False)
    return ((x.float() - as_t(mean)) / as_t(std)).to(rgb.dtype)


# ------------------------------------------------------------------------------------------------------------
# CLIP text tower — `text_encoder` slot of MarigoldPipeline (marigold_pipeline.py:147-153); only ever fed the empty prompt
# (marigold_pipeline.py:356-369: 2 tokens; training/train.py:455-458: padded to 77), once per process.
SD2_TEXT = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=23, num_attention_heads=16, max_position_embeddings=77,
                vocab_size=49408, hidden_act="gelu", layer_norm_eps=1e-5)   # stabilityai/stable-diffusion-2 text_encoder (OpenCLIP ViT-H text, penultimate layer)
BOS_ID, EOS_ID = 49406, 49407


def empty_prompt_ids(padding="do_not_pad", max_length=77, pad_token_id=0):
    """token ids of tokenizer("") without needing the vocabulary files: <|startoftext|> <|endoftext|>, then (padding="max_length")
    the pad token — "!" = id 0 in the SD-2 tokenizer (special_tokens_map.json), <|endoftext|> = 49407 in SD-1.x / openai CLIP."""
    ids = [BOS_ID, EOS_ID]
    if padding == "max_length":
        ids = ids + [pad_token_id] * (max_length - 2)
    elif padding != "do_not_pad":
        raise ValueError(padding)
    return torch.tensor([ids], dtype=torch.int64)


class _TextOut:
    def __init__(self, last_hidden_state):
        self.last_hidden_state = last_hidden_state

    def __getitem__(self, i):
        return (self.last_hidden_state,)[i]


class CLIPCausalAttention(CLIPAttention):
    def forward(self, x, residual):
        C = x.shape[-1]
        bias = F.cached(self, "bqkv_%s" % x.dtype, (self.q_proj.bias, self.k_proj.bias, self.v_proj.bias),
                        lambda: torch.cat([self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]).detach().to(x.dtype))
        qkv = F.linear(x, (self.q_proj.weight, self.k_proj.weight, self.v_proj.weight), bias, owner=self, name="wqkv")
        # <= 77 tokens, once per process: GEMM + masked row softmax + GEMM (the fused kernel has no mask input)
        a = F._attn_core_unfused(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], self.heads, self.scale, causal=True)
        return self.out_proj(a, residual=residual)


class _TextEmbeddings(nn.Module):
    def __init__(self, vocab, positions, dim):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, dim)
        self.position_embedding = nn.Embedding(positions, dim)

    def forward(self, input_ids):
        L = input_ids.shape[1]
        return self.token_embedding.weight[input_ids] + self.position_embedding.weight[:L]   # a 77-row gather: torch (plumbing)


class CLIPTextTransformer(nn.Module):
    def __init__(self, c):
        super().__init__()
        dim = c["hidden_size"]
        self.embeddings = _TextEmbeddings(c["vocab_size"], c["max_position_embeddings"], dim)
        self.encoder = _Encoder(c["num_hidden_layers"], dim, c["intermediate_size"], c["num_attention_heads"], c["hidden_act"], c["layer_norm_eps"])
        for layer in self.encoder.layers:
            att = CLIPCausalAttention(dim, c["num_attention_heads"])
            layer.self_attn = att
        self.final_layer_norm = LayerNorm(dim, eps=c["layer_norm_eps"])

    def forward(self, input_ids):
        x = self.embeddings(input_ids).contiguous()
        for layer in self.encoder.layers:
            x = layer(x)
        return self.final_layer_norm(x)


class CLIPTextModel(_HubIO, nn.Module):
    """`text_encoder(input_ids)[0]` -> last_hidden_state [B, L, C] (after final_layer_norm), transformers state-dict layout."""
    _class_name, _prefix, _config_section, _defaults = "CLIPTextModel", "text_model.", "text_config", SD2_TEXT

    def __init__(self, **kwargs):
        super().__init__()
        cfg = dict(SD2_TEXT)
        cfg.update(kwargs)
        self.config = cfg
        self.text_model = CLIPTextTransformer(cfg)

    @property
    def dtype(self):
        return self.text_model.final_layer_norm.weight.dtype

    @property
    def device(self):
        return self.text_model.final_layer_norm.weight.device

    def load_state_dict(self, sd, strict=True, **kw):
        sd = {k: v for k, v in sd.items() if not k.endswith("position_ids")}
        return super().load_state_dict(sd, strict=strict, **kw)

    @torch.no_grad()
    @ops.device_scoped
    def forward(self, input_ids, return_dict=True, **unused):
        out = self.text_model(input_ids.to(self.device))
        return _TextOut(out) if return_dict else (out,)
