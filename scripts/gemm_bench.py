"""Micro-benchmark of one GEMM shape (out[M,N] = a[M,K] w[N,K]^T + bias) through the C ABI, for A/B runs of kernel heuristics.
usage: python scripts/gemm_bench.py M N K [iters=20] [dtype=fp16] [residual=0] [option=value ...]   (options: scripts/_options.py, e.g. persistent=0)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_e2e_ft_amd import ops
import _options

a = _options.take(sys.argv[1:])
M, N, K = (int(v) for v in a[:3])
iters = int(a[3]) if len(a) > 3 else 20
dt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[a[4] if len(a) > 4 else "fp16"]
res = int(a[5]) if len(a) > 5 else 0
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((M, K), generator=g, device=dev).to(dt)
w = (torch.randn((N, K), generator=g, device=dev) / K ** 0.5).to(dt)
b = torch.randn((N,), generator=g, device=dev).to(dt)
r = torch.randn((M, N), generator=g, device=dev).to(dt) if res else None
out = torch.empty((M, N), dtype=dt, device=dev)
for _ in range(3):
    ops.gemm(x, w, b, r, out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    ops.gemm(x, w, b, r, out)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
print("gemm M%d N%d K%d %s res%d: %.3f ms  %.1f TFLOP/s" % (M, N, K, a[4] if len(a) > 4 else "fp16", res, ms, 2.0 * M * N * K / ms / 1e9))
