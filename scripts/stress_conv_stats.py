"""Stress test for the GroupNorm statistics emitted by the implicit-GEMM epilogue (ops.conv2d(..., gn_stats=True)).

Repeats a convolution whose grid puts two 4-wave workgroups on some CUs (B=3, 48x48, Cout=640 -> 270 tiles of 128x128 on
256 CUs) from an idle GPU and checks every launch's partial statistics against the mean of the tensor it wrote.  This is the
reproducer for the gfx950 packed-fp32 operand-swizzle glitch described in DESIGN.md §3.6 (before the igemm files were built with
-fno-slp-vectorize it reported 20-25 bad launches out of 60 on the first shape).

    python scripts/stress_conv_stats.py [iters]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_e2e_ft_amd import ops

dev = torch.device("cuda")
torch.manual_seed(0)


def check(B, H, W, cin, cout, k, res, idle, iters, dtype=torch.float16):
    x = torch.randn(B, H, W, cin, device=dev, dtype=dtype)
    w = (torch.randn(cout, k * k * cin, device=dev) / (k * k * cin) ** 0.5).to(dtype)
    b = torch.randn(cout, device=dev).to(dtype)
    r = torch.randn(B, H, W, cout, device=dev, dtype=dtype) if res else None
    bad, worst, first = 0, 0.0, None
    for _ in range(iters):
        if idle:
            torch.cuda.synchronize()
            time.sleep(0.02)
        o = ops.conv2d(x, w, b, cout, k, k, 1, (k // 2,) * 4, residual=r, gn_stats=True)
        st = o._e2eft_gn
        p = st.partial.view(B, st.nslabs, cout, 3)
        err = (p[..., 1] - o.float().view(B, st.nslabs, -1, cout).mean(2)).abs().max().item()
        same = first is None or (torch.equal(first[0], o) and torch.equal(first[1], st.partial))
        if first is None:
            first = (o.clone(), st.partial.clone())
        worst = max(worst, err)
        bad += int(err > 2e-3 or not same)
    print("B%d %dx%d %d->%d k%d res%d %s: bad %d/%d worst |mean err| %.3g nslabs %d" %
          (B, H, W, cin, cout, k, res, "idle" if idle else "back-to-back", bad, iters, worst, st.nslabs), flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    total = 0
    for idle in (True, False):
        total += check(3, 48, 48, 1920, 640, 3, False, idle, n)
        total += check(3, 48, 48, 640, 640, 3, True, idle, n)
    sys.exit(1 if total else 0)
