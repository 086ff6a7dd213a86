"""GroupNorm+SiLU -> conv3x3 as two passes against the fused route (e2eft_conv2d_fwd_normed).  usage: python scripts/norm_conv_bench.py B H W Cin Cout [iters=20]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_e2e_ft_amd import ops
import _options

a = _options.take(sys.argv[1:])
B, H, W, Ci, Co = (int(v) for v in a[:5])
iters = int(a[5]) if len(a) > 5 else 20
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((B, H, W, Ci), generator=g, device=dev).half()
w = (torch.randn((Co, 9 * Ci), generator=g, device=dev) / (9 * Ci) ** 0.5).half()
b = torch.randn((Co,), generator=g, device=dev).half()
ga, be = torch.ones(Ci, device=dev).half(), torch.zeros(Ci, device=dev).half()
x = ops.conv2d(x, torch.eye(Ci, device=dev).half().repeat_interleave(1, 0).reshape(Ci, Ci), None, Ci, 1, 1, 1, (0, 0, 0, 0), gn_stats=True)   # a producer: statistics attached


def two():
    h = ops.groupnorm(x, ga, be, 32, 1e-5, silu=True)
    return ops.conv2d(h, w, b, Co, 3, 3, 1, (1, 1, 1, 1), gn_stats=True)


def one():
    return ops.conv2d(x, w, b, Co, 3, 3, 1, (1, 1, 1, 1), gn_stats=True, norm=(ga, be, 32, 1e-5, True))


for name, fn in (("two passes", two), ("fused", one), ("two passes", two), ("fused", one)):
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        y = fn()
    e.record()
    torch.cuda.synchronize()
    print("%s B%d %dx%d %d->%d: %.3f ms (fused route taken: %s)" % (name, B, H, W, Ci, Co, s.elapsed_time(e) / iters, getattr(y, "_e2eft_keep", None) is not None))
