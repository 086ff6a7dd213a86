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


TINY_TEXT = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, max_position_embeddings=77,
                 vocab_size=49408, hidden_act="gelu", layer_norm_eps=1e-5)


def tiny_text_sd(seed=13, cfg=TINY_TEXT):
    from diffusion_e2e_ft_amd.clip import CLIPTextModel
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in CLIPTextModel(**cfg).state_dict().items():
        if "layer_norm" in k and k.endswith("weight"):
            sd[k] = 1 + 0.05 * torch.randn(v.shape, generator=g)
        elif "embedding" in k:
            sd[k] = 0.3 * torch.randn(v.shape, generator=g)
        elif v.dim() >= 2:
            sd[k] = torch.randn(v.shape, generator=g) / v[0].numel() ** 0.5
        else:
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
    return sd


def test_text_oracle_matches_transformers_and_layout():
    tr = pytest.importorskip("transformers")
    from diffusion_e2e_ft_amd.clip import CLIPTextModel, empty_prompt_ids, SD2_TEXT
    cfg = tr.CLIPTextConfig(**TINY_TEXT, pad_token_id=1, bos_token_id=49406, eos_token_id=49407)
    ref = tr.CLIPTextModel(cfg).eval()
    # hub checkpoints (and transformers 4.x, which the reference pins) prefix every key with "text_model."; transformers 5 dropped
    # the wrapper level in memory and re-maps on load.  The product module keeps the checkpoint layout.
    flat = not next(iter(ref.state_dict())).startswith("text_model.")
    pre = "text_model." if flat else ""
    a = {pre + k: tuple(v.shape) for k, v in ref.state_dict().items() if not k.endswith("position_ids")}
    b = {k: tuple(v.shape) for k, v in CLIPTextModel(**TINY_TEXT).state_dict().items()}
    assert a == b
    sd = tiny_text_sd()
    res = ref.load_state_dict({(k[len("text_model."):] if flat else k): v for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys and not [k for k in res.missing_keys if not k.endswith("position_ids")]
    for padding in ("do_not_pad", "max_length"):
        ids = empty_prompt_ids(padding)
        assert ids.shape == (1, 2 if padding == "do_not_pad" else 77) and ids[0, 0] == 49406 and ids[0, 1] == 49407
        with torch.no_grad():
            want = ref(input_ids=ids)[0]
            got = clip_ref.clip_text_ref(sd, TINY_TEXT, ids)
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-5), padding
    H, I, L = SD2_TEXT["hidden_size"], SD2_TEXT["intermediate_size"], SD2_TEXT["num_hidden_layers"]
    per_layer = 4 * (H * H + H) + (H * I + I) + (I * H + H) + 4 * H
    assert 49408 * H + 77 * H + L * per_layer + 2 * H == 340_387_840   # SD-2 text_encoder (OpenCLIP ViT-H text, 23 layers)
