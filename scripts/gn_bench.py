"""GroupNorm(+SiLU) micro-benchmark: is the apply pass HBM-bound or transcendental-bound?  usage: python scripts/gn_bench.py [B H W C]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_e2e_ft_amd import ops

a = [int(v) for v in sys.argv[1:]]
B, H, W, C = (a + [8, 768, 768, 128][len(a):])[:4]
dev = torch.device("cuda")
for dt in (torch.float16, torch.float32):
    x = torch.randn((B, H, W, C), device=dev, dtype=dt)
    g, b = torch.ones(C, device=dev, dtype=dt), torch.zeros(C, device=dev, dtype=dt)
    out = torch.empty_like(x)
    for silu in (False, True):
        for _ in range(3):
            ops.groupnorm(x, g, b, 32, 1e-6, silu=silu, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.groupnorm(x, g, b, 32, 1e-6, silu=silu, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        nbytes = x.numel() * x.element_size()
        print("gn %s B%d %dx%d C%d silu=%d: %.3f ms (3 passes: stats read + apply read/write = %.2f TB/s)" % (str(dt).split(".")[1], B, H, W, C, silu, ms, 3 * nbytes / ms / 1e9))
