"""igemm5.hip (persistent workgroups walking a tile sequence, wave-private sliced epilogue) against torch CPU.  The variant takes only
problems with at least two tiles per CU; e2eft_set_option(E2EFT_OPT_PERSISTENT_GRID, 8) shrinks the grid to eight workgroups so that SMALL problems — which the
CPU reference finishes in seconds — run through it: every operand mode of the FAST path (3x3 / strided / 1x1-as-GEMM / fused upsample /
two-source concat / zero-insertion dgrad), GEMM and batched GEMM, 3, 4, 5 and many k-tiles per tile, ragged N tiles, tile counts that
do not divide by the grid, several images per tile, every epilogue option, the fused GroupNorm statistics.  The cases run in a
subprocess (so that the option never leaks into other tests of the session); the debug counter proves the variant really ran."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

SCRIPT = r'''
import sys, os
sys.path.insert(0, os.path.join(%r, ".."))
sys.path.insert(0, %r)
import ctypes
import torch
import torch.nn.functional as F
from diffusion_e2e_ft_amd import ops, _lib
from util import nhwc, to_nchw, pack_conv_weight, q, rel_err, TOL
dev = torch.device("cuda:0")
lib = _lib.load()
_lib.set_option(_lib.OPT_PERSISTENT_GRID, int(os.environ.get("TEST_PERSISTENT_GRID", "0")))
_lib.set_option(_lib.OPT_PERSISTENT, int(os.environ.get("TEST_PERSISTENT", "1")))
_lib.set_option(_lib.OPT_PATCH_CONV, 0)      # this file is about igemm5: the halo-patch kernel (tests/test_patch_conv_gpu.py) would take the 3x3 cases
_lib.set_option(_lib.OPT_PERSISTENT_MIN_QROUNDS, 8)   # the cases below were written against "two tiles per workgroup"; round 6's default is half a round (own test at the end of the file)
lib.e2eft_debug_persistent_launches.restype = ctypes.c_long
EXPECT = int(os.environ.get("EXPECT_PERSISTENT", "1"))
def launches():
    return lib.e2eft_debug_persistent_launches()
worst = 0.0
# ---- convolutions: B, H, W, C1, C2, Co, k, stride, up_to, rowadd, residual, alpha     (B*Hout*Wout must be a multiple of 256)
cases = [
    (2, 64, 64, 64, 0, 128, 3, 1, None, False, False, 1.0),      # 32 tiles, 9 k-tiles
    (1, 64, 64, 128, 0, 128, 3, 1, None, True, True, 0.7),       # 16 tiles (two per workgroup), rowadd + residual + alpha
    (1, 96, 96, 64, 0, 320, 3, 1, None, False, True, 1.0),       # 36 x 3 tiles, ragged last N tile (64 wide), 108 = 8 * 13 + 4
    (3, 32, 32, 64, 128, 192, 3, 1, None, True, False, 1.0),     # two sources, source switch inside the k sequence, 12 x 2 tiles
    (8, 16, 16, 192, 0, 128, 3, 1, None, False, False, 1.0),     # one image per tile row block: 8 images = 8 tiles x 1 -> N 128: 8 tiles (< 16: not eligible)
    (16, 16, 16, 192, 0, 256, 3, 1, None, False, False, 1.0),    # one 256-row tile = one image; 32 tiles
    (64, 8, 8, 64, 0, 128, 3, 1, None, False, True, 1.0),        # four images per tile
    (2, 64, 64, 64, 0, 128, 3, 2, None, False, False, 1.0),      # stride 2: 2 * 32 * 32 = 2048 rows, 8 tiles -> not eligible; kept as a fall-through check
    (4, 64, 64, 64, 0, 256, 3, 2, None, True, False, 1.0),       # stride 2, 16 x 2 tiles
    (1, 32, 32, 128, 0, 128, 3, 1, (64, 64), False, False, 1.0), # fused nearest upsample 32 -> 64
    (1, 64, 64, 192, 0, 128, 1, 1, None, False, True, 1.0),      # 1x1 -> GEMM mode, exactly 3 k-tiles
    (1, 64, 64, 256, 0, 136, 1, 1, None, True, False, 1.0),      # GEMM mode, 4 k-tiles, N = 136 (ragged: 8 columns in the 2nd tile)
    (1, 64, 64, 320, 0, 128, 1, 1, None, False, False, 1.0),     # 5 k-tiles
]
for dtype in (torch.float16, torch.bfloat16):
    for (B, H, W, C1, C2, Co, k, st, up, ra, rs, alpha) in cases:
        g = torch.Generator().manual_seed(B * 1000 + H * 10 + C1 + Co + k)
        x = q(torch.randn(B, C1, H, W, generator=g), dtype)
        x2 = q(torch.randn(B, C2, H, W, generator=g), dtype) if C2 else None
        w = q(torch.randn(Co, C1 + C2, k, k, generator=g) / ((C1 + C2) * k * k) ** 0.5, dtype)
        b = q(torch.randn(Co, generator=g), dtype)
        xin = x if x2 is None else torch.cat([x, x2], dim=1)
        if up is not None:
            xin = F.interpolate(xin, size=up, mode="nearest")
        pd = k // 2
        ref = F.conv2d(xin.double(), w.double(), b.double(), stride=st, padding=pd).float()
        rav = q(torch.randn(B, Co, generator=g), dtype) if ra else None
        rsv = q(torch.randn(ref.shape, generator=g), dtype) if rs else None
        if rav is not None:
            ref = ref + rav[:, :, None, None]
        ref = ref * alpha
        if rsv is not None:
            ref = ref + rsv
        n0 = launches()
        out = ops.conv2d(nhwc(x, dtype, dev), pack_conv_weight(w, dtype, dev), b.to(dtype).to(dev), Co, k, k, st, (pd, pd, pd, pd),
                         x2=None if x2 is None else nhwc(x2, dtype, dev), up_to=up, rowadd=None if rav is None else rav.to(dtype).to(dev),
                         residual=None if rsv is None else nhwc(rsv, dtype, dev), alpha=alpha)
        torch.cuda.synchronize()
        took = launches() - n0
        tiles = (ref.shape[0] * ref.shape[2] * ref.shape[3] // 256) * ((Co + 127) // 128)
        e = rel_err(to_nchw(out), ref)
        ok = e <= TOL[dtype] and bool(torch.isfinite(out.float()).all())
        print("%%s conv %%s rel err %%.2e persistent=%%d tiles=%%d %%s" %% (str(dtype)[6:], (B, H, W, C1, C2, Co, k, st, up), e, took, tiles, "ok" if ok else "FAIL"))
        worst = max(worst, e / TOL[dtype])
        assert ok
        if EXPECT:
            assert took == (1 if tiles >= 16 else 0), (took, tiles)
        else:
            assert took == 0

# ---- zero-insertion dgrad of a strided convolution (the training path's conv2d_dgrad)
for dtype in (torch.float16,):
    g = torch.Generator().manual_seed(5)
    B, Ci, Co, H = 4, 64, 128, 64
    w = q(torch.randn(Co, Ci, 3, 3, generator=g) / 24.0, dtype)
    dy = q(torch.randn(B, Co, H // 2, H // 2, generator=g), dtype)
    xr = torch.zeros(B, Ci, H, H, dtype=torch.double, requires_grad=True)
    yr = F.conv2d(xr, w.double(), None, stride=2, padding=1)
    yr.backward(dy.double())
    wd = w.permute(1, 2, 3, 0).flip(1, 2).reshape(Ci, 9 * Co).contiguous().to(dtype).to(dev)   # [Ci, (ky, kx, co)] flipped taps
    n0 = launches()
    dx = ops.conv2d_dgrad(nhwc(dy, dtype, dev), wd, (B, H, H, Ci), 0, 3, 3, 2, (1, 1, 1, 1), None, 1.0)
    torch.cuda.synchronize()
    e = rel_err(to_nchw(dx if not isinstance(dx, tuple) else dx[0]), xr.grad.float())
    print("dgrad stride 2 rel err %%.2e persistent=%%d" %% (e, launches() - n0))
    assert e <= TOL[dtype]

# ---- GEMM / batched GEMM
for dtype in (torch.float16, torch.bfloat16):
    for (M, N, K, bias, res, alpha) in [(4096, 320, 320, True, True, 1.0), (2048, 512, 1024, False, False, 0.5), (8192, 64, 192, True, False, 1.0), (1024, 1280, 640, True, True, 1.0)]:
        g = torch.Generator().manual_seed(M + N + K)
        a = q(torch.randn(M, K, generator=g), dtype)
        w = q(torch.randn(N, K, generator=g) / K ** 0.5, dtype)
        b = q(torch.randn(N, generator=g), dtype) if bias else None
        r = q(torch.randn(M, N, generator=g), dtype) if res else None
        ref = (a.double() @ w.double().t())
        if b is not None:
            ref = ref + b.double()
        ref = ref * alpha
        if r is not None:
            ref = ref + r.double()
        n0 = launches()
        out = ops.gemm(a.to(dtype).to(dev), w.to(dtype).to(dev), None if b is None else b.to(dtype).to(dev), None if r is None else r.to(dtype).to(dev), alpha=alpha)
        torch.cuda.synchronize()
        took = launches() - n0
        e = rel_err(out.float().cpu(), ref.float())
        tiles = (M // 256) * ((N + 127) // 128)
        print("%%s gemm %%s rel err %%.2e persistent=%%d tiles=%%d" %% (str(dtype)[6:], (M, N, K), e, took, tiles))
        worst = max(worst, e / TOL[dtype])
        assert e <= TOL[dtype]
        if EXPECT:
            assert took == (1 if tiles >= 16 else 0), (took, tiles)
    # batched: z = 2 x 3, [M, K] x [N, K]^T per z with strides
    zo, zi, M, N, K = 2, 3, 512, 256, 256
    g = torch.Generator().manual_seed(77)
    a = q(torch.randn(zo, zi, M, K, generator=g), dtype)
    w = q(torch.randn(zo, zi, N, K, generator=g) / 16.0, dtype)
    ref = torch.einsum("abmk,abnk->abmn", a.double(), w.double())
    ad, wd = a.to(dtype).to(dev), w.to(dtype).to(dev)
    out = torch.empty(zo, zi, M, N, dtype=dtype, device=dev)
    n0 = launches()
    ops.bgemm_raw(dtype, M, N, K, ad, K, (zi * M * K, M * K), wd, K, (zi * N * K, N * K), out, N, (zi * M * N, M * N), zo, zi, alpha=1.0)
    torch.cuda.synchronize()
    e = rel_err(out.float().cpu(), ref.float())
    print("%%s bgemm rel err %%.2e persistent=%%d" %% (str(dtype)[6:], e, launches() - n0))
    assert e <= TOL[dtype]
    if EXPECT:
        assert launches() - n0 == 1

# ---- fused GroupNorm statistics: the consumer GroupNorm gives the same result with the producer's partials as with its own pass
# (B, H, C1, Co, residual, rowadd, alpha, stride): the last two rows add the per-image row vector + alpha and a strided producer
for (B, H, C1, Co, res, ra, alpha, st) in [(4, 32, 64, 128, False, False, 1.0, 1), (1, 64, 128, 320, True, False, 1.0, 1), (16, 16, 64, 64, True, False, 1.0, 1),
                                           (2, 64, 64, 128, True, True, 0.5, 1), (4, 64, 64, 256, False, True, 1.0, 2)]:
    g = torch.Generator().manual_seed(B + H + Co + st)
    x = q(torch.randn(B, C1, H, H, generator=g) * 3.0 + 1.5, torch.float16)
    w = q(torch.randn(Co, C1, 3, 3, generator=g) / (C1 * 9) ** 0.5, torch.float16)
    bb = q(torch.randn(Co, generator=g) * 4.0, torch.float16)
    Ho = H // st
    r = nhwc(q(torch.randn(B, Co, Ho, Ho, generator=g) * 2.0 - 5.0, torch.float16), torch.float16, dev) if res else None
    rav = q(torch.randn(B, Co, generator=g) * 2.0, torch.float16).half().to(dev) if ra else None
    n0 = launches()
    y = ops.conv2d(nhwc(x, torch.float16, dev), pack_conv_weight(w, torch.float16, dev), bb.half().to(dev), Co, 3, 3, st, (1, 1, 1, 1), residual=r, rowadd=rav,
                   alpha=alpha, gn_stats=True)
    took = launches() - n0
    yref = F.conv2d(x.double(), w.double(), bb.double(), stride=st, padding=1)
    if ra:
        yref = yref + rav.double().cpu()[:, :, None, None]
    yref = yref * alpha
    if res:
        yref = yref + to_nchw(r).double()
    assert rel_err(to_nchw(y), yref.float()) <= TOL[torch.float16]
    assert getattr(y, "_e2eft_gn", None) is not None, "no GroupNorm statistics emitted"
    ga, be = torch.ones(Co, device=dev).half(), torch.zeros(Co, device=dev).half()
    a = ops.groupnorm(y, ga, be, 32, 1e-5, True)
    b_ = ops.groupnorm(y.clone(), ga, be, 32, 1e-5, True)     # no statistics attached: the norm computes its own
    ref = F.silu(F.group_norm(to_nchw(y).double(), 32, eps=1e-5)).float()
    e1, e2 = rel_err(to_nchw(a), ref), rel_err(to_nchw(b_), ref)
    print("gn stats %%s: with partials %%.2e, own pass %%.2e persistent=%%d" %% ((B, H, C1, Co, res, ra, alpha, st), e1, e2, took))
    assert e1 < 2e-3 and e2 < 2e-3
    if EXPECT:
        assert took == 1

# ---- determinism: the same launch twice, bit-identical
x = nhwc(q(torch.randn(2, 64, 64, 64), torch.float16), torch.float16, dev)
w = pack_conv_weight(q(torch.randn(128, 64, 3, 3) / 24.0, torch.float16), torch.float16, dev)
y1 = ops.conv2d(x, w, None, 128, 3, 3, 1, (1, 1, 1, 1), gn_stats=True)
y2 = ops.conv2d(x, w, None, 128, 3, 3, 1, (1, 1, 1, 1), gn_stats=True)
assert torch.equal(y1, y2) and torch.equal(y1._e2eft_gn.partial, y2._e2eft_gn.partial)
print("PERSISTENT CASES PASSED worst %%.2f of tolerance" %% worst)
''' % (HERE, HERE)


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, timeout=900)
    print(r.stdout[-6000:])
    assert r.returncode == 0 and "PERSISTENT CASES PASSED" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


def test_persistent_kernel_on_small_shapes(dev):
    _run({"TEST_PERSISTENT_GRID": "8", "EXPECT_PERSISTENT": "1"})


def test_same_cases_without_the_variant(dev):
    """E2EFT_OPT_PERSISTENT = 0: every case runs on igemm2 — the reference numbers of the A/B"""
    _run({"TEST_PERSISTENT": "0", "EXPECT_PERSISTENT": "0"})


def test_half_a_round_of_tiles_is_enough_since_round_6(dev):
    """E2EFT_OPT_PERSISTENT_MIN_QROUNDS (default 2 = half a round of the machine, was two rounds): a launch whose tile count lies between half a round and two
    rounds runs on the persistent kernel — one tile per workgroup, some workgroups idle: its k-loop (barrier under the last MFMA groups, 1.3-1.45 k cycles per
    k-tile) is what pays, not the tile overlap (profiles/r06c_min_rounds_sweep.txt: GEMM 18432 x 640 x 640 + residual 467 -> 568 TF/s) — with results equal to
    igemm2's up to the summation order; below half a round it still falls through."""
    import ctypes
    import torch
    import torch.nn.functional as F
    from diffusion_e2e_ft_amd import ops, _lib
    from util import nhwc, to_nchw, pack_conv_weight, q, rel_err, TOL
    lib = _lib.load()
    lib.e2eft_debug_persistent_launches.restype = ctypes.c_long
    dtype = torch.float16
    assert lib.e2eft_get_option(_lib.OPT_PERSISTENT_MIN_QROUNDS) == 2
    for (B, H, W, Ci, Co, expect) in [(8, 16, 16, 192, 128, 1), (4, 16, 16, 192, 128, 1), (3, 16, 16, 192, 128, 0)]:     # 8 / 4 / 3 tiles on the 8-workgroup test grid
        g = torch.Generator().manual_seed(B)
        x = q(torch.randn(B, Ci, H, W, generator=g), dtype)
        w = q(torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5, dtype)
        b = q(torch.randn(Co, generator=g), dtype)
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).float()
        with _lib.option(_lib.OPT_PERSISTENT_GRID, 8), _lib.option(_lib.OPT_PATCH_CONV, 0):
            n0 = lib.e2eft_debug_persistent_launches()
            y = ops.conv2d(nhwc(x, dtype, dev), pack_conv_weight(w, dtype, dev), b.to(dtype).to(dev), Co, 3, 3, 1, (1, 1, 1, 1))
            took = lib.e2eft_debug_persistent_launches() - n0
        torch.cuda.synchronize()
        assert took == expect, (B, took)
        assert rel_err(to_nchw(y), ref) <= TOL[dtype]
