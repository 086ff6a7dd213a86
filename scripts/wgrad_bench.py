"""Micro-benchmark of the direct weight-gradient kernel (csrc/wgrad.hip) through the C ABI.  usage: python scripts/wgrad_bench.py B H W Cin Cout [k=3] [iters=10] [dtype=bf16|fp16|fp32]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_e2e_ft_amd import ops

B, H, W, Ci, Co = (int(v) for v in sys.argv[1:6])
k = int(sys.argv[6]) if len(sys.argv) > 6 else 3
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 10
dn = sys.argv[8] if len(sys.argv) > 8 else "bf16"
dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[dn]
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((B, H, W, Ci), generator=g, device=dev).to(dt)
dy = torch.randn((B, H, W, Co), generator=g, device=dev).to(dt)
p = k // 2
for _ in range(2):
    dw = ops.conv2d_wgrad(dy, x, None, Co, k, k, 1, (p, p, p, p), 1.0)
assert dw is not None
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    dw = ops.conv2d_wgrad(dy, x, None, Co, k, k, 1, (p, p, p, p), 1.0)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
print("wgrad %dx%d B%d %dx%d %d->%d %s: %.3f ms  %.1f TFLOP/s" % (k, k, B, H, W, Ci, Co, dn, ms, 2.0 * B * H * W * Co * k * k * Ci / ms / 1e9))
