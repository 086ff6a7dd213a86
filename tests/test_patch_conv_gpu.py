"""igemm6.hip (persistent 3x3 convolution whose A operand is a 2-D halo patch in LDS) against torch CPU fp64.  As for igemm5
(tests/test_persistent_gpu.py) e2eft_set_option(E2EFT_OPT_PERSISTENT_GRID, 8) sends SMALL problems through it: 2, 3 and 5 channel chunks,
two-source concat with the switch on and off a chunk boundary, one tile row / one tile column per image, tile counts that do not divide
by the grid, ragged N tiles, every epilogue option (bias / per-image row vector / alpha / residual), both dtypes, the fused GroupNorm
statistics, the stride-1 dgrad, and shapes that are NOT eligible (width not a multiple of 32) which must fall through to igemm5 with
equal results.  Runs in a subprocess so that the options never leak; a debug counter proves which kernel ran."""
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
_lib.set_option(_lib.OPT_PATCH_CONV, int(os.environ.get("TEST_PATCH_CONV", "1")))
_lib.set_option(_lib.OPT_PERSISTENT_MIN_QROUNDS, 8)   # the eligibility expectations below are "two tiles per workgroup" (round 6's default is half a round: tests/test_persistent_gpu.py)
lib.e2eft_debug_patch_launches.restype = ctypes.c_long
EXPECT = int(os.environ.get("TEST_PATCH_CONV", "1"))
def launches():
    return lib.e2eft_debug_patch_launches()
worst = 0.0
# B, H, W, C1, C2, Co, rowadd, residual, alpha, eligible[, fused nearest upsample to]
cases = [
    (2, 16, 32, 128, 0, 128, False, False, 1.0, True, (32, 64)),   # fused 2x upsample, 16 tiles
    (1, 32, 32, 192, 0, 256, True, True, 1.0, True, (64, 64)),     # fused 2x upsample, 3 chunks, 16 x 2 tiles, rowadd + residual
    (2, 16, 32, 128, 0, 128, False, False, 1.0, False, (48, 96)),  # 3x upsample: not eligible

    (2, 32, 64, 128, 0, 128, False, False, 1.0, True),     # 16 tiles, 2 chunks
    (1, 64, 64, 192, 0, 128, True, True, 0.7, True),       # 16 tiles, 3 chunks, rowadd + residual + alpha
    (1, 96, 96, 64, 64, 320, False, True, 1.0, True),      # 36 x 3 tiles (108 = 8 * 13 + 4), ragged last N tile, concat switch on the chunk boundary
    (3, 32, 32, 64, 128, 192, True, False, 1.0, True),     # one tile column per image, concat 64 + 128 (3 chunks), 12 x 2 tiles
    (10, 32, 32, 320, 0, 136, False, False, 1.0, True),    # 5 chunks, N = 136 (8 columns in the 2nd N tile), 40 x 2 tiles (enough 128-row tiles that the host does not split K)
    (4, 16, 32, 320, 0, 136, False, False, 1.0, False),    # the same layer small: the host splits K by taps, not eligible
    (1, 8, 512, 128, 0, 128, False, True, 1.0, True),      # one tile row per image (top and bottom padding in every tile)
    (17, 8, 32, 128, 0, 128, True, False, 1.0, True),      # one tile = one image, 17 tiles: the per-XCD chunks are ragged
    (2, 48, 48, 128, 0, 128, False, False, 1.0, False),    # width 48: not eligible
    (2, 36, 64, 128, 0, 128, False, False, 1.0, False),    # height 36: not eligible
    (2, 32, 64, 64, 0, 128, False, False, 1.0, False),     # one chunk: not eligible
]
for dtype in (torch.float16, torch.bfloat16):
    for case in cases:
        (B, H, W, C1, C2, Co, ra, rs, alpha, elig), up = case[:10], (case[10] if len(case) > 10 else None)
        g = torch.Generator().manual_seed(B * 1000 + H * 10 + C1 + Co + W)
        x = q(torch.randn(B, C1, H, W, generator=g), dtype)
        x2 = q(torch.randn(B, C2, H, W, generator=g), dtype) if C2 else None
        w = q(torch.randn(Co, C1 + C2, 3, 3, generator=g) / ((C1 + C2) * 9) ** 0.5, dtype)
        b = q(torch.randn(Co, generator=g), dtype)
        xin = x if x2 is None else torch.cat([x, x2], dim=1)
        if up is not None:
            xin = F.interpolate(xin, size=up, mode="nearest")
        ref = F.conv2d(xin.double(), w.double(), b.double(), stride=1, padding=1).float()
        rav = q(torch.randn(B, Co, generator=g), dtype) if ra else None
        rsv = q(torch.randn(ref.shape, generator=g), dtype) if rs else None
        if rav is not None:
            ref = ref + rav[:, :, None, None]
        ref = ref * alpha
        if rsv is not None:
            ref = ref + rsv
        n0 = launches()
        out = ops.conv2d(nhwc(x, dtype, dev), pack_conv_weight(w, dtype, dev), b.to(dtype).to(dev), Co, 3, 3, 1, (1, 1, 1, 1),
                         x2=None if x2 is None else nhwc(x2, dtype, dev), up_to=up, rowadd=None if rav is None else rav.to(dtype).to(dev),
                         residual=None if rsv is None else nhwc(rsv, dtype, dev), alpha=alpha)
        torch.cuda.synchronize()
        took = launches() - n0
        tiles = (ref.shape[0] * ref.shape[2] * ref.shape[3] // 256) * ((Co + 127) // 128)
        e = rel_err(to_nchw(out), ref)
        ok = e <= TOL[dtype] and bool(torch.isfinite(out.float()).all())
        print("%%s conv %%s rel err %%.2e patch=%%d tiles=%%d %%s" %% (str(dtype)[6:], (B, H, W, C1, C2, Co, up), e, took, tiles, "ok" if ok else "FAIL"), flush=True)
        worst = max(worst, e / TOL[dtype])
        assert ok
        assert took == (1 if (EXPECT and elig and tiles >= 16) else 0), (took, tiles)

# ---- a single nonzero input pixel / a single nonzero tap: the halo geometry exactly (every output pixel of a 3 x 3 neighbourhood, across tile borders)
for (py, px) in [(0, 0), (7, 31), (8, 32), (15, 63), (31, 0), (16, 33)]:
    x = torch.zeros(2, 128, 32, 64)
    x[1, 5, py, px] = 1.0
    x[0, 77, 31 - py, 63 - px] = -2.0
    w = torch.arange(9, dtype=torch.float32).reshape(1, 1, 3, 3).add(1.0).repeat(128, 128, 1, 1) / 16.0
    ref = F.conv2d(x.double(), w.double(), None, padding=1).float()
    n0 = launches()
    out = ops.conv2d(nhwc(x, torch.float16, dev), pack_conv_weight(w, torch.float16, dev), None, 128, 3, 3, 1, (1, 1, 1, 1))
    torch.cuda.synchronize()
    assert torch.equal(to_nchw(out).float(), ref), (py, px)
    assert launches() - n0 == EXPECT
print("impulse responses exact", flush=True)

# ---- stride-1 dgrad (the training path's conv2d_dgrad with flipped taps): an eligible 3x3 problem of its own
for dtype in (torch.float16,):
    g = torch.Generator().manual_seed(5)
    B, Ci, Co, H, W = 2, 128, 192, 32, 64
    w = q(torch.randn(Co, Ci, 3, 3, generator=g) / 24.0, dtype)
    dy = q(torch.randn(B, Co, H, W, generator=g), dtype)
    xr = torch.zeros(B, Ci, H, W, dtype=torch.double, requires_grad=True)
    yr = F.conv2d(xr, w.double(), None, stride=1, padding=1)
    yr.backward(dy.double())
    wd = w.permute(1, 2, 3, 0).flip(1, 2).reshape(Ci, 9 * Co).contiguous().to(dtype).to(dev)
    n0 = launches()
    dx = ops.conv2d_dgrad(nhwc(dy, dtype, dev), wd, (B, H, W, Ci), 0, 3, 3, 1, (1, 1, 1, 1), None, 1.0)
    torch.cuda.synchronize()
    e = rel_err(to_nchw(dx if not isinstance(dx, tuple) else dx[0]), xr.grad.float())
    print("dgrad stride 1 rel err %%.2e patch=%%d" %% (e, launches() - n0), flush=True)
    assert e <= TOL[dtype]
    assert launches() - n0 == EXPECT

# ---- fused GroupNorm statistics
for (B, H, W, C1, Co, res, ra, alpha) in [(4, 32, 32, 128, 128, False, False, 1.0), (1, 64, 64, 128, 320, True, False, 1.0), (2, 32, 64, 192, 128, True, True, 0.5)]:
    g = torch.Generator().manual_seed(B + H + Co)
    x = q(torch.randn(B, C1, H, W, generator=g) * 3.0 + 1.5, torch.float16)
    w = q(torch.randn(Co, C1, 3, 3, generator=g) / (C1 * 9) ** 0.5, torch.float16)
    bb = q(torch.randn(Co, generator=g) * 4.0, torch.float16)
    r = nhwc(q(torch.randn(B, Co, H, W, generator=g) * 2.0 - 5.0, torch.float16), torch.float16, dev) if res else None
    rav = q(torch.randn(B, Co, generator=g) * 2.0, torch.float16).half().to(dev) if ra else None
    n0 = launches()
    y = ops.conv2d(nhwc(x, torch.float16, dev), pack_conv_weight(w, torch.float16, dev), bb.half().to(dev), Co, 3, 3, 1, (1, 1, 1, 1), residual=r, rowadd=rav,
                   alpha=alpha, gn_stats=True)
    took = launches() - n0
    yref = F.conv2d(x.double(), w.double(), bb.double(), stride=1, padding=1)
    if ra:
        yref = yref + rav.double().cpu()[:, :, None, None]
    yref = yref * alpha
    if res:
        yref = yref + to_nchw(r).double()
    assert rel_err(to_nchw(y), yref.float()) <= TOL[torch.float16]
    assert getattr(y, "_e2eft_gn", None) is not None, "no GroupNorm statistics emitted"
    ga, be = torch.ones(Co, device=dev).half(), torch.zeros(Co, device=dev).half()
    a = ops.groupnorm(y, ga, be, 32, 1e-5, True)
    b_ = ops.groupnorm(y.clone(), ga, be, 32, 1e-5, True)
    ref = F.silu(F.group_norm(to_nchw(y).double(), 32, eps=1e-5)).float()
    e1, e2 = rel_err(to_nchw(a), ref), rel_err(to_nchw(b_), ref)
    print("gn stats %%s: with partials %%.2e, own pass %%.2e patch=%%d" %% ((B, H, W, C1, Co, res, ra, alpha), e1, e2, took), flush=True)
    assert e1 < 2e-3 and e2 < 2e-3
    assert took == EXPECT

# ---- determinism: the same launch twice, bit-identical
x = nhwc(q(torch.randn(2, 128, 32, 64), torch.float16), torch.float16, dev)
w = pack_conv_weight(q(torch.randn(128, 128, 3, 3) / 34.0, torch.float16), torch.float16, dev)
y1 = ops.conv2d(x, w, None, 128, 3, 3, 1, (1, 1, 1, 1), gn_stats=True)
y2 = ops.conv2d(x, w, None, 128, 3, 3, 1, (1, 1, 1, 1), gn_stats=True)
assert torch.equal(y1, y2) and torch.equal(y1._e2eft_gn.partial, y2._e2eft_gn.partial)
print("PATCH CASES PASSED worst %%.2f of tolerance" %% worst)
''' % (HERE, HERE)


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, timeout=600)
    print(r.stdout[-6000:])
    assert r.returncode == 0 and "PATCH CASES PASSED" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


def test_patch_kernel_on_small_shapes(dev):
    _run({"TEST_PERSISTENT_GRID": "8", "TEST_PATCH_CONV": "1"})


def test_same_cases_on_igemm5(dev):
    """E2EFT_OPT_PATCH_CONV = 0: the same cases on igemm5 — the reference numbers of the A/B"""
    _run({"TEST_PERSISTENT_GRID": "8", "TEST_PATCH_CONV": "0"})
