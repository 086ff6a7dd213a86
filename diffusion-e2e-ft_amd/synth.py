"""Seeded synthetic initialisation of product modules (no checkpoints exist offline — SURVEY.md §8c/§8d).
Variance-preserving: conv / linear ~ N(0, 1/fan_in), residual-branch output layers x0.5, norm gamma = 1 + 0.1 N,
beta = 0.1 N, biases 0.05 N.  Used by bench.py / smoke for random-init weights of the right architecture; parity tests
load oracle-generated state dicts instead."""
import torch

_RESIDUAL_OUT = (".conv2.weight", ".to_out.0.weight", ".ff.net.2.weight", ".proj_out.weight")


@torch.no_grad()
def init_synthetic_(module, seed=1234):
    dev = next(module.parameters()).device
    g = torch.Generator(device=dev).manual_seed(seed)
    for name, p in module.named_parameters():
        is_norm = ".norm" in name or "group_norm" in name or "conv_norm_out" in name or "layer_norm" in name or "layernorm" in name or "layrnorm" in name
        if name.endswith(".weight") and p.dim() >= 2:
            fan_in = p[0].numel()
            w = torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32) * (1.0 / fan_in) ** 0.5
            if name.endswith(_RESIDUAL_OUT):
                w *= 0.5
        elif name.endswith(".weight") and is_norm:
            w = 1.0 + 0.1 * torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32)
        elif is_norm:
            w = 0.1 * torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32)
        else:
            w = 0.05 * torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32)
        p.copy_(w.to(p.dtype))
    return module
