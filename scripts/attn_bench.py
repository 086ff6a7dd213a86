"""Micro-benchmark of the fused attention forward (d = 64) through the C ABI.  usage: python scripts/attn_bench.py B heads N [iters=20] [dtype=fp16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_e2e_ft_amd import ops
import _options

a = _options.take(sys.argv[1:])
B, H, N = (int(v) for v in a[:3])
iters = int(a[3]) if len(a) > 3 else 20
dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[a[4] if len(a) > 4 else "fp16"]
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn((B, N, 3 * H * 64), generator=g, device=dev).to(dt)
C = H * 64
q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
for _ in range(3):
    o = ops.attention(q, k, v, H, 0.125)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    o = ops.attention(q, k, v, H, 0.125)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
print("attn B%d h%d N%d %s: %.3f ms  %.1f TFLOP/s  checksum %.6f" % (B, H, N, a[4] if len(a) > 4 else "fp16", ms, 4.0 * B * H * N * N * 64 / ms / 1e9, o.float().abs().mean().item()))
