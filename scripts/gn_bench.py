"""GroupNorm(+SiLU) micro-benchmark through the C ABI.  usage: python scripts/gn_bench.py [B H W C] [pre=1] [option=value ...]
pre=1: the statistics come from a producer (a 1x1 convolution's epilogue), so the op is finalize + apply as in the inference path; pre=0: partial + finalize + apply."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_e2e_ft_amd import ops
import _options

a = [int(v) for v in _options.take(sys.argv[1:])]
B, H, W, C = (a + [8, 768, 768, 128][len(a):])[:4]
pre = a[4] if len(a) > 4 else 1
dev = torch.device("cuda")
dt = torch.float16
if pre:
    xin = torch.randn((B, H, W, 64), device=dev, dtype=dt)
    w = (torch.randn((C, 64), device=dev) / 8).to(dt)
    x = ops.conv2d(xin, w, None, C, 1, 1, 1, (0, 0, 0, 0), gn_stats=True)
    assert getattr(x, "_e2eft_gn", None) is not None
else:
    x = torch.randn((B, H, W, C), device=dev, dtype=dt)
g, b = torch.ones(C, device=dev, dtype=dt), torch.zeros(C, device=dev, dtype=dt)
out = torch.empty_like(x)
for silu in (True,):
    for _ in range(5):
        ops.groupnorm(x, g, b, 32, 1e-6, silu=silu, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.groupnorm(x, g, b, 32, 1e-6, silu=silu, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    nbytes = x.numel() * x.element_size()
    print("gn fp16 B%d %dx%d C%d silu=%d pre=%d iters=%d: %.1f us  (read + write = %.2f TB/s)" % (B, H, W, C, silu, pre, ops._lib.load().e2eft_get_option(13), us, 2 * nbytes / us / 1e6))
