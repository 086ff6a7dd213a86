"""Does a GroupNorm apply run faster when the tensor it reads was written just before and fits the 256 MB Infinity Cache?
Per-image time of (conv 128->128 @768^2 with fused statistics) + (GroupNorm + SiLU) for batch 1, 2, 4, 8."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_e2e_ft_amd import ops

dev = torch.device("cuda")
C = 128
w = (torch.randn((C, 9 * C), device=dev) / (9 * C) ** 0.5).half()
b = torch.randn((C,), device=dev).half()
ga, be = torch.ones(C, device=dev).half(), torch.zeros(C, device=dev).half()
for B in (1, 2, 4, 8):
    x = torch.randn((B, 768, 768, C), device=dev).half()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    tc = tg = 0.0
    it = 10
    for i in range(it + 2):
        ev[0].record()
        y = ops.conv2d(x, w, b, C, 3, 3, 1, (1, 1, 1, 1), gn_stats=True)
        ev[1].record()
        z = ops.groupnorm(y, ga, be, 32, 1e-5, True)
        ev[2].record()
        torch.cuda.synchronize()
        if i >= 2:
            tc += ev[0].elapsed_time(ev[1]); tg += ev[1].elapsed_time(ev[2])
    print("B=%d: conv %.3f ms/image, groupnorm %.3f ms/image (%.2f TB/s algorithmic)" % (B, tc / it / B, tg / it / B, 2 * 768 * 768 * C * 2 * B / (tg / it) / 1e9))
