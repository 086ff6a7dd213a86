"""Micro-benchmark of the fused d = 512 attention (VAE mid block) through the C ABI.  usage: python scripts/attn512_bench.py B N [iters=10] [dtype=fp16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_e2e_ft_amd import ops

B, N = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[sys.argv[4] if len(sys.argv) > 4 else "fp16"]
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn((B, N, 1536), generator=g, device=dev).to(dt)
q, k, v = qkv[..., :512], qkv[..., 512:1024], qkv[..., 1024:]
for _ in range(2):
    o = ops.attention512(q, k, v, 512 ** -0.5)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    o = ops.attention512(q, k, v, 512 ** -0.5)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
print("attn512 B%d N%d: %.3f ms  %.1f TFLOP/s" % (B, N, ms, 4.0 * B * N * N * 512 / ms / 1e9))
