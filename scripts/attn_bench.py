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
dt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[a[4] if len(a) > 4 else "fp16"]
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
if len(a) > 5 and a[5] == "bwd":      # fused backward (dq, dk, dv from q, k, v, out, dout and the forward's lse): 14 B h N^2 64 flops
    o, lse = ops.attention(q, k, v, H, 0.125, return_lse=True)
    do = torch.randn(o.shape, generator=g, device=dev).to(dt)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:]
    for _ in range(2):
        ops.attention_bwd(q, k, v, o, do, lse, H, 0.125, dq, dk, dv)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        ops.attention_bwd(q, k, v, o, do, lse, H, 0.125, dq, dk, dv)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print("attn_bwd B%d h%d N%d %s: %.3f ms  %.1f TFLOP/s" % (B, H, N, a[4], ms, 14.0 * B * H * N * N * 64 / ms / 1e9))
