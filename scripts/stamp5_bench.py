"""Per-tile phase clocks of the persistent implicit-GEMM kernel (igemm5.hip, instrumented build).
usage: python -m diffusion_e2e_ft_amd.build --stamps
       E2EFT_LIB=diffusion-e2e-ft_amd/lib/libe2eft_stamps.so python scripts/stamp5_bench.py B H W Cin Cout [k=3] [residual=0] [stats=1]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diffusion_e2e_ft_amd import ops, _lib

a = sys.argv[1:]
B, H, W, Ci, Co = (int(v) for v in a[:5])
k = int(a[5]) if len(a) > 5 else 3
res = int(a[6]) if len(a) > 6 else 0
stats = int(a[7]) if len(a) > 7 else 1
dev = torch.device("cuda")
x = torch.randn((B, H, W, Ci), device=dev).half()
w = (torch.randn((Co, k * k * Ci), device=dev) / (k * k * Ci) ** 0.5).half()
b = torch.randn((Co,), device=dev).half()
r = torch.randn((B, H, W, Co), device=dev).half() if res else None
p = k // 2
for _ in range(int(os.environ.get("WARM", "20"))):
    out = ops.conv2d(x, w, b, Co, k, k, 1, (p, p, p, p), residual=r, gn_stats=bool(stats))
torch.cuda.synchronize()
lib = _lib.load()
nwg = 256
buf = (ctypes.c_longlong * (nwg * 32 * 8))()
lib.e2eft_debug_read_stamps5.restype = ctypes.c_int
assert lib.e2eft_debug_read_stamps5(buf, nwg) == 0
s = np.frombuffer(buf, dtype=np.int64).reshape(nwg, 32, 8)
ntile = ((B * H * W) // 256) * ((Co + 127) // 128)
per = min(32, ntile // 256)
nk = k * k * Ci // 64
t = s[:, 1:per - 1, :]            # steady tiles (not the first, not the last)
print("conv%dx%d B%d %dx%d %d->%d res=%d stats=%d: %d tiles, %d per workgroup, %d k-tiles" % (k, k, B, H, W, Ci, Co, res, stats, ntile, ntile // 256, nk))
print("steady k-tiles (0 .. nk-3)        mean %8.0f cycles = %.0f / k-tile" % ((t[:, :, 1] - t[:, :, 0]).mean(), (t[:, :, 1] - t[:, :, 0]).mean() / (nk - 2)))
print("last two k-tiles (switch + last)  mean %8.0f cycles = switch %.0f + last %.0f" % ((t[:, :, 2] - t[:, :, 1]).mean(), (t[:, :, 4] - t[:, :, 1]).mean(), (t[:, :, 2] - t[:, :, 4]).mean()))
print("epilogue                          mean %8.0f cycles = slice 0 %.0f + slices 1-3 %.0f + slices 4-7 and statistics %.0f" % ((t[:, :, 3] - t[:, :, 2]).mean(), (t[:, :, 5] - t[:, :, 2]).mean(), (t[:, :, 6] - t[:, :, 5]).mean(), (t[:, :, 3] - t[:, :, 6]).mean()))
print("barrier + statistics merge        mean %8.0f cycles" % (t[:, :, 7] - t[:, :, 3]).mean())
nxt = s[:, 2:per, 0] - s[:, 1:per - 1, 3]
print("epilogue exit -> next k-loop      mean %8.0f cycles" % nxt.mean())
print("tile period                       mean %8.0f cycles" % (s[:, 2:per, 0] - s[:, 1:per - 1, 0]).mean())
print("first tile: k-loop %.0f, last two %.0f, epilogue %.0f" % ((s[:, 0, 1] - s[:, 0, 0]).mean(), (s[:, 0, 2] - s[:, 0, 1]).mean(), (s[:, 0, 3] - s[:, 0, 2]).mean()))
