"""Time the test-time ensembling kernels on a full-size stack (default 10 x 768 x 768 fp32) and the whole ensemble_depths call.
usage: python scripts/ensemble_bench.py [N=10] [H=768] [W=768]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_e2e_ft_amd import ops
from diffusion_e2e_ft_amd.ensemble import ensemble_depths, ensemble_normals

a = [int(v) for v in sys.argv[1:]]
N, H, W = (a + [10, 768, 768][len(a):])[:3]
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand((N, H, W), generator=g, device=dev) * (1 + torch.arange(N, device=dev).view(-1, 1, 1) * 0.1)
s = torch.ones(N, device=dev)
t = torch.zeros(N, device=dev)
xn = torch.randn((N, 3, H, W), generator=g, device=dev)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


P = H * W
rows = [("minmax", lambda: ops.ensemble_minmax(x), 4 * N * P),
        ("gram", lambda: ops.ensemble_gram(x), 4 * N * P),                       # algorithmic: one read of the stack (re-reads hit L2 / MALL)
        ("depth_reduce (objective, no images)", lambda: ops.ensemble_depth_reduce(x, s, t, want_images=False), 4 * N * P),
        ("depth_reduce (median + MAD)", lambda: ops.ensemble_depth_reduce(x, s, t), 4 * (N + 2) * P),
        ("depth_reduce (mean + std)", lambda: ops.ensemble_depth_reduce(x, s, t, use_mean=True), 4 * (N + 2) * P),
        ("normals", lambda: ops.ensemble_normals(xn), 24 * N * P)]
for name, fn, nbytes in rows:
    us = timeit(fn)
    print("%-40s %8.1f us  %7.1f GB/s (algorithmic %.1f MB)" % (name, us, nbytes / us / 1e3, nbytes / 1e6))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for _ in range(2):
        ensemble_depths(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ensemble_depths(x)
    torch.cuda.synchronize()
    print("ensemble_depths(N=%d, %dx%d) end to end: %.2f ms (1 + 2N objective evaluations through scipy)" % (N, H, W, (time.perf_counter() - t0) / 5 * 1e3))
    t0 = time.perf_counter()
    for _ in range(5):
        ensemble_normals(xn)
    torch.cuda.synchronize()
    print("ensemble_normals end to end: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
