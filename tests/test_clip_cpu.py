"""CPU pin of oracle/clip_ref.py against the transformers CLIP vision tower installed in this image (the third-party module the
reference calls, geowizard_pipeline.py:232-248), plus structural checks of the product module's state-dict layout."""
import pytest
import torch

from oracle import clip_ref

TINY = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=56, patch_size=14,
            projection_dim=48, hidden_act="quick_gelu", layer_norm_eps=1e-5, num_channels=3)


def tiny_clip_sd(seed=11, cfg=TINY):
    from diffusion_e2e_ft_amd.clip import CLIPVisionModelWithProjection
    g = torch.Generator().manual_seed(seed)
    m = CLIPVisionModelWithProjection(**cfg)
    sd = {}
    for k, v in m.state_dict().items():
        if k.endswith("norm.weight") or k.endswith("norm1.weight") or k.endswith("norm2.weight"):
            sd[k] = 1 + 0.05 * torch.randn(v.shape, generator=g)
        elif v.dim() >= 2:
            fan_in = v[0].numel()
            sd[k] = torch.randn(v.shape, generator=g) / fan_in ** 0.5
        else:
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
    sd["vision_model.embeddings.position_embedding.weight"] = 0.1 * torch.randn(sd["vision_model.embeddings.position_embedding.weight"].shape, generator=g)
    return sd


def test_state_dict_layout_matches_transformers():
    tr = pytest.importorskip("transformers")
    from diffusion_e2e_ft_amd.clip import CLIPVisionModelWithProjection
    ref = tr.CLIPVisionModelWithProjection(tr.CLIPVisionConfig(**TINY))
    mine = CLIPVisionModelWithProjection(**TINY)
    a = {k: tuple(v.shape) for k, v in ref.state_dict().items() if not k.endswith("position_ids")}
    b = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
    assert a == b
    from diffusion_e2e_ft_amd.clip import CLIP_VIT_L14 as c   # the full-size config, by formula (no allocation)
    H, I = c["hidden_size"], c["intermediate_size"]
    per_layer = 4 * (H * H + H) + (H * I + I) + (I * H + H) + 4 * H
    total = H + 3 * 14 * 14 * H + 257 * H + 2 * H + c["num_hidden_layers"] * per_layer + 2 * H + H * c["projection_dim"]
    assert total == 303_966_208, total   # openai/clip-vit-large-patch14 vision tower (303.18 M) + 1024x768 projection


def test_oracle_matches_transformers():
    tr = pytest.importorskip("transformers")
    sd = tiny_clip_sd()
    ref = tr.CLIPVisionModelWithProjection(tr.CLIPVisionConfig(**TINY)).eval()
    missing = ref.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if not k.endswith("position_ids")] and not missing.unexpected_keys
    x = torch.randn(2, 3, 56, 56, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        out = ref(pixel_values=x)
        emb, last = clip_ref.clip_vision_ref(sd, TINY, x)
    assert torch.allclose(emb, out.image_embeds, rtol=1e-5, atol=1e-6)
    assert torch.allclose(last, out.last_hidden_state, rtol=1e-5, atol=1e-5)
