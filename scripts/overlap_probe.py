"""Can an HBM-bound GroupNorm run UNDER a power-bound persistent GEMM on a second stream?  The GEMM-mode igemm5 kernel allocates 192 registers per
lane and wave (two waves per SIMD), which leaves room for gn_apply waves (64 registers); the conv-mode kernel (240) does not.
Prints the time of N GEMMs alone, N GroupNorms alone, and both loops issued concurrently on two streams."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_e2e_ft_amd import ops

dev = torch.device("cuda")
M, N, K = 73728, 2560, 1280
a = torch.randn((M, K), device=dev).half()
w = (torch.randn((N, K), device=dev) / K ** 0.5).half()
out = torch.empty((M, N), device=dev, dtype=torch.float16)
x = torch.randn((8, 384, 384, 256), device=dev).half()
ga, be = torch.ones(256, device=dev).half(), torch.zeros(256, device=dev).half()
y = torch.empty_like(x)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
it = 20


def gemms():
    for _ in range(it):
        ops.gemm(a, w, out=out)


def gns():
    for _ in range(3 * it):
        ops.groupnorm(x, ga, be, 32, 1e-5, True, out=y)


def timed(f1, f2):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if f1:
        with torch.cuda.stream(s1):
            s1.wait_event(e0)
            f1()
    if f2:
        with torch.cuda.stream(s2):
            s2.wait_event(e0)
            f2()
    torch.cuda.current_stream().wait_stream(s1)
    torch.cuda.current_stream().wait_stream(s2)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for _ in range(2):
    timed(gemms, gns)
tg, tn, tb = timed(gemms, None), timed(None, gns), timed(gemms, gns)
print("GEMM loop alone %.2f ms (%.0f TFLOP/s); GroupNorm loop alone %.2f ms; both on two streams %.2f ms (sum %.2f, max %.2f)" % (
    tg, 2.0 * M * N * K * it / tg / 1e9, tn, tb, tg + tn, max(tg, tn)))
