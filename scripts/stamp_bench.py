"""Per-workgroup phase clocks of igemm2 (instrumented build, -DE2EFT_STAMPS; DESIGN.md §3 'fixed cost per workgroup').
usage: python -m diffusion_e2e_ft_amd.build --stamps
       E2EFT_LIB=diffusion-e2e-ft_amd/lib/libe2eft_stamps.so [NOSTATS=1] python scripts/stamp_bench.py B H W Cin Cout [k=3]"""
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
dev = torch.device("cuda")
x = torch.randn((B, H, W, Ci), device=dev).half()
w = (torch.randn((Co, k * k * Ci), device=dev) / (k * k * Ci) ** 0.5).half()
b = torch.randn((Co,), device=dev).half()
p = k // 2
warm = int(os.environ.get("WARM", "3"))      # WARM=400: the stamps are those of a launch after ~1 s of the same kernel (clocks settled)
for _ in range(warm):
    out = ops.conv2d(x, w, b, Co, k, k, 1, (p, p, p, p), gn_stats=os.environ.get("NOSTATS") is None)
torch.cuda.synchronize()
nwg = min(65536, ((B * H * W + 255) // 256) * ((Co + 127) // 128))
buf = (ctypes.c_longlong * (nwg * 8))()
lib = _lib.load()
lib.e2eft_debug_read_stamps.restype = ctypes.c_int
rc = lib.e2eft_debug_read_stamps(buf, nwg)
assert rc == 0, rc
s = np.frombuffer(buf, dtype=np.int64).reshape(nwg, 8)
rt_reader = lib.e2eft_debug_read_stamps_rt
if not s[:, 4].any():       # the launch went to the row-strip kernel (igemm4.hip keeps its own stamp arrays)
    lib.e2eft_debug_read_stamps4.restype = ctypes.c_int
    assert lib.e2eft_debug_read_stamps4(buf, nwg) == 0
    s = np.frombuffer(buf, dtype=np.int64).reshape(nwg, 8).copy()
    s[:, 5:8] = s[:, 1:2]
    rt_reader = lib.e2eft_debug_read_stamps4_rt
    print("(row-strip kernel igemm4)")
names = ["prologue (start -> loop entry)", "main loop", "accumulators -> LDS + barrier", "epilogue stores (+GN stats)"]
for i, n in enumerate(names):
    d = s[:, i + 1] - s[:, i]
    print("%-34s mean %8.0f  median %8.0f cycles" % (n, d.mean(), np.median(d)))
print("k-loop start-up: tile 0 %.0f cycles, tiles 1-2 %.0f, tiles 3-5 %.0f (loop entry -> after tile 0 / 2 / 5)" % (
    (s[:, 5] - s[:, 1]).mean(), (s[:, 6] - s[:, 5]).mean(), (s[:, 7] - s[:, 6]).mean()))
tot = s[:, 4] - s[:, 0]
print("%-34s mean %8.0f cycles; k-tiles %d -> %.0f cycles / k-tile in the loop" % ("workgroup total", tot.mean(), k * k * Ci // 64, (s[:, 2] - s[:, 1]).mean() / (k * k * Ci // 64)))

# shader clock actually delivered while this kernel ran, from the kernel's own two counters (no profiler attached): s_memtime cycles per
# s_memrealtime tick (100 MHz) between the first and the last stamp of every workgroup
rt = (ctypes.c_longlong * (nwg * 2))()
rt_reader.restype = ctypes.c_int
if rt_reader(rt, nwg) == 0:
    r = np.frombuffer(rt, dtype=np.int64).reshape(nwg, 2)
    ticks = (r[:, 1] - r[:, 0]).astype(np.float64)
    ok = ticks > 0
    mhz = tot[ok] / ticks[ok] * 100.0
    print("shader clock during the kernel (s_memtime / s_memrealtime): mean %.0f MHz, median %.0f, p5 %.0f, p95 %.0f; workgroup wall time mean %.1f us"
          % (mhz.mean(), np.median(mhz), np.percentile(mhz, 5), np.percentile(mhz, 95), ticks[ok].mean() / 100.0))

if rt_reader is not lib.e2eft_debug_read_stamps_rt:
    n4 = min(nwg, 4096)
    tb = (ctypes.c_longlong * (n4 * 64))()
    lib.e2eft_debug_read_tile4.restype = ctypes.c_int
    if lib.e2eft_debug_read_tile4(tb, n4) == 0:
        t = np.frombuffer(tb, dtype=np.int64).reshape(n4, 8, 8)
        t = t[:, t[0, :, 5] != 0, :]                              # the waves the kernel has (4)
        d = np.diff(t[:, :, :6], axis=2).astype(np.float64)      # [wg, wave, 5 phases]
        names = ["DMA wait (s_waitcnt vmcnt)", "barrier", "first two fragment groups read (+ A DMA issue)", "first MFMA group", "remaining 3 groups + reads + DMA issue"]
        print("inside one steady-state k-tile (tile 9), per wave, mean over waves and workgroups / mean of the per-workgroup MAX over waves:")
        for i, n in enumerate(names):
            print("  %-44s %7.0f / %7.0f cycles" % (n, d[:, :, i].mean(), d[:, :, i].max(axis=1).mean()))
        print("  %-44s %7.0f cycles" % ("whole tile", (t[:, :, 5] - t[:, :, 0]).mean()))
