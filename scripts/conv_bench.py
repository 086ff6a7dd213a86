"""Micro-benchmark of one implicit-GEMM conv shape through the C ABI (for rocprofv3 PMC passes and A/B runs).
usage: python scripts/conv_bench.py B H W Cin Cout [k=3] [iters=10] [dtype=fp16] [residual=0] [option=value ...]   (options: scripts/_options.py, e.g. persistent=0)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_e2e_ft_amd import ops
import _options

a = _options.take(sys.argv[1:])
B, H, W, Ci, Co = (int(v) for v in a[:5])
k = int(a[5]) if len(a) > 5 else 3
iters = int(a[6]) if len(a) > 6 else 10
dt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[a[7] if len(a) > 7 else "fp16"]
res = int(a[8]) if len(a) > 8 else 0
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((B, H, W, Ci), generator=g, device=dev).to(dt)
w = (torch.randn((Co, k * k * Ci), generator=g, device=dev) / (k * k * Ci) ** 0.5).to(dt)
b = torch.randn((Co,), generator=g, device=dev).to(dt)
r = torch.randn((B, H, W, Co), generator=g, device=dev).to(dt) if res else None
out = torch.empty((B, H, W, Co), dtype=dt, device=dev)
p = k // 2
for _ in range(3):
    ops.conv2d(x, w, b, Co, k, k, 1, (p, p, p, p), residual=r, out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    ops.conv2d(x, w, b, Co, k, k, 1, (p, p, p, p), residual=r, out=out)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
fl = 2.0 * B * H * W * Co * k * k * Ci
print("conv%dx%d B%d %dx%d %d->%d %s: %.3f ms  %.1f TFLOP/s" % (k, k, B, H, W, Ci, Co, a[7] if len(a) > 7 else "fp16", ms, fl / ms / 1e9))
