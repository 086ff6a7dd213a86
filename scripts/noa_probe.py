"""EXPERIMENT (instrumented library only: `python -m diffusion_e2e_ft_amd.build --stamps`, E2EFT_LIB=.../libe2eft_stamps.so): what would an A-operand
reuse scheme buy the persistent implicit-GEMM kernel at most?  The kernel is run with its A-operand LDS-DMA pieces (4 of the 6 pieces a wave issues
per k-tile, 32 of the 48 KB a workgroup moves L2 -> LDS) issued always / never / on the first tap of a filter row only (what a row-strip reuse
needs) / on the first tap of a 64-channel chunk only (a 2-D halo patch in LDS serving all nine taps).  Results are WRONG by construction (stale LDS
data); time and clock are what is measured.  usage: E2EFT_LIB=... python scripts/noa_probe.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_e2e_ft_amd import ops, _lib

lib = _lib.load()
dev = torch.device("cuda")
dt = torch.float16
shapes = [(8, 768, 768, 128, 128), (8, 384, 384, 256, 256), (8, 192, 192, 512, 512)]
for (B, H, W, Ci, Co) in shapes:
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((B, H, W, Ci), generator=g, device=dev).to(dt)
    w = (torch.randn((Co, 9 * Ci), generator=g, device=dev) / (9 * Ci) ** 0.5).to(dt)
    b = torch.randn((Co,), generator=g, device=dev).to(dt)
    out = torch.empty((B, H, W, Co), dtype=dt, device=dev)
    fl = 2.0 * B * H * W * Co * 9 * Ci
    for flags, name in ((0, "A pieces always (production)"), (2, "A pieces on kx = 0 only (1/3)"), (4, "A pieces on the first tap only (1/9)"), (1, "A pieces never"), (0, "always, again")):
        lib.e2eft_debug_set_flags5(flags)
        for _ in range(30):
            ops.conv2d(x, w, b, Co, 3, 3, 1, (1, 1, 1, 1), out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(40):
            ops.conv2d(x, w, b, Co, 3, 3, 1, (1, 1, 1, 1), out=out)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 40
        print("conv3x3 B%d %dx%d %d->%d  %-40s %.3f ms  %.1f TFLOP/s" % (B, H, W, Ci, Co, name, ms, fl / ms / 1e9), flush=True)
    lib.e2eft_debug_set_flags5(0)
