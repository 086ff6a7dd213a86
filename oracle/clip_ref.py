"""TEST INFRASTRUCTURE (see oracle/__init__.py).  CPU restatement of the CLIP ViT image encoder behind GeoWizard's image
conditioning (/root/reference/GeoWizard/geowizard/models/geowizard_pipeline.py:232-248 calls
`transformers.CLIPVisionModelWithProjection(...).image_embeds`; transformers is a third-party dependency of the reference,
requirements pin 4.x, not vendored).  PINNED: tests/test_clip_cpu.py checks this restatement against the transformers
implementation installed in this image on a seeded small config (same state dict, same input)."""
import torch
import torch.nn.functional as F


def clip_vision_ref(sd, cfg, pixel_values):
    """functional forward from a transformers-layout state dict -> (image_embeds [B, proj], last_hidden_state [B, 1+g*g, C])"""
    p = "vision_model."
    C, heads, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"]
    x = F.conv2d(pixel_values, sd[p + "embeddings.patch_embedding.weight"], stride=cfg["patch_size"])      # [B,C,g,g]
    x = x.flatten(2).transpose(1, 2)
    cls = sd[p + "embeddings.class_embedding"].expand(x.shape[0], 1, C)
    x = torch.cat([cls, x], 1) + sd[p + "embeddings.position_embedding.weight"][None]
    ln = lambda t, k: F.layer_norm(t, (C,), sd[k + ".weight"], sd[k + ".bias"], eps)
    lin = lambda t, k: F.linear(t, sd[k + ".weight"], sd.get(k + ".bias"))
    x = ln(x, p + "pre_layrnorm")
    for i in range(cfg["num_hidden_layers"]):
        k = p + "encoder.layers.%d." % i
        h = ln(x, k + "layer_norm1")
        B, N, _ = h.shape
        split = lambda t: t.view(B, N, heads, C // heads).transpose(1, 2)
        a = F.scaled_dot_product_attention(split(lin(h, k + "self_attn.q_proj")), split(lin(h, k + "self_attn.k_proj")), split(lin(h, k + "self_attn.v_proj")))
        x = x + lin(a.transpose(1, 2).reshape(B, N, C), k + "self_attn.out_proj")
        h = lin(ln(x, k + "layer_norm2"), k + "mlp.fc1")
        if cfg["hidden_act"] == "quick_gelu":
            h = h * torch.sigmoid(1.702 * h)
        elif cfg["hidden_act"] == "gelu":
            h = F.gelu(h)
        else:
            raise ValueError(cfg["hidden_act"])
        x = x + lin(h, k + "mlp.fc2")
    pooled = ln(x[:, 0], p + "post_layernorm")
    return F.linear(pooled, sd["visual_projection.weight"]), x


def clip_text_ref(sd, cfg, input_ids):
    """transformers CLIPTextModel forward (causal self-attention, pre-LN blocks, final_layer_norm) -> last_hidden_state [B, L, C]"""
    p = "text_model."
    C, heads, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"]
    B, L = input_ids.shape
    x = sd[p + "embeddings.token_embedding.weight"][input_ids] + sd[p + "embeddings.position_embedding.weight"][:L][None]
    ln = lambda t, k: F.layer_norm(t, (C,), sd[k + ".weight"], sd[k + ".bias"], eps)
    lin = lambda t, k: F.linear(t, sd[k + ".weight"], sd.get(k + ".bias"))
    for i in range(cfg["num_hidden_layers"]):
        k = p + "encoder.layers.%d." % i
        h = ln(x, k + "layer_norm1")
        split = lambda t: t.view(B, L, heads, C // heads).transpose(1, 2)
        a = F.scaled_dot_product_attention(split(lin(h, k + "self_attn.q_proj")), split(lin(h, k + "self_attn.k_proj")), split(lin(h, k + "self_attn.v_proj")), is_causal=True)
        x = x + lin(a.transpose(1, 2).reshape(B, L, C), k + "self_attn.out_proj")
        h = lin(ln(x, k + "layer_norm2"), k + "mlp.fc1")
        h = h * torch.sigmoid(1.702 * h) if cfg["hidden_act"] == "quick_gelu" else F.gelu(h)
        x = x + lin(h, k + "mlp.fc2")
    return ln(x, p + "final_layer_norm")
