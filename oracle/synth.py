"""Seeded synthetic checkpoints in diffusers key layout (no weights exist offline — SURVEY.md §8c/§8d).
TEST INFRASTRUCTURE ONLY.  Variance-preserving init so that activations stay O(1) through 60+ layers:
conv / linear ~ N(0, 1/fan_in); norm gamma = 1 + 0.1 N, beta = 0.1 N; biases 0.05 N; the last layer of every
residual branch (conv2, to_out.0, ff.net.2, proj_out) scaled by 0.5."""
import zlib

import torch

_RESIDUAL_OUT = (".conv2.weight", ".to_out.0.weight", ".ff.net.2.weight", ".proj_out.weight")


def synth_state_dict(shapes, seed=1234, dtype=torch.float32):
    sd = {}
    for key in sorted(shapes):
        shape = shapes[key]
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 31))
        is_norm = ".norm" in key or "group_norm" in key or key.startswith("conv_norm_out") or ".conv_norm_out" in key
        if key.endswith(".weight") and len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            w = torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5
            if key.endswith(_RESIDUAL_OUT):
                w = w * 0.5
            sd[key] = w.to(dtype)
        elif key.endswith(".weight") and is_norm:
            sd[key] = (1.0 + 0.1 * torch.randn(shape, generator=g)).to(dtype)
        elif key.endswith(".bias") and is_norm:
            sd[key] = (0.1 * torch.randn(shape, generator=g)).to(dtype)
        else:
            sd[key] = (0.05 * torch.randn(shape, generator=g)).to(dtype)
    return sd


def synth_inputs(batch, height, width, ctx_len, ctx_dim, seed=0):
    """Seeded synthetic inputs mirroring marigold_pipeline.py:245 (uint8 image -> [-1,1]) and an empty-prompt-like
    context tensor."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (batch, 3, height, width), generator=g, dtype=torch.int64).to(torch.uint8)
    rgb = img.float() / 255.0 * 2.0 - 1.0
    ctx = 0.5 * torch.randn((1, ctx_len, ctx_dim), generator=g)
    return rgb, ctx


def fast_state_dict(shapes, seed=1234, dtype=torch.float32):
    """Cheap stand-in weights for TIMING the oracle only (bench.py cpu_baseline): one 1M-value random block tiled into
    every tensor with the variance-preserving scale; ~100x faster to build than synth_state_dict at SD size."""
    g = torch.Generator().manual_seed(seed)
    block = torch.randn(1 << 20, generator=g)
    sd = {}
    for key, shape in shapes.items():
        n = 1
        for s_ in shape:
            n *= s_
        reps = (n + block.numel() - 1) // block.numel()
        flat = block.repeat(reps)[:n] if reps > 1 else block[:n].clone()
        if key.endswith(".weight") and len(shape) >= 2:
            fan_in = n // shape[0]
            flat = flat * (1.0 / fan_in) ** 0.5
        elif key.endswith(".weight"):
            flat = 1.0 + 0.1 * flat
        else:
            flat = 0.05 * flat
        sd[key] = flat.reshape(shape).to(dtype)
    return sd
