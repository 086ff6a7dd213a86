"""convin.hip (3x3 / stride-1 / pad-1 convolution with EIGHT input channels, operands straight from global memory, persistent workgroups)
against torch CPU fp64: one and several N tiles, ragged N, with and without bias, both dtypes, tile counts that do not divide by the grid, one
tile row per image, the fused GroupNorm statistics, impulse responses across tile borders (bit-exact), shapes that must fall through
(width not a multiple of 32; a residual).  Subprocess + 8-workgroup grid as tests/test_patch_conv_gpu.py; a debug counter proves which kernel ran."""
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
_lib.set_option(_lib.OPT_PERSISTENT_GRID, 8)
_lib.set_option(_lib.OPT_THIN_INPUT_CONV, int(os.environ.get("TEST_THIN", "1")))
lib.e2eft_debug_thin_launches.restype = ctypes.c_long
EXPECT = int(os.environ.get("TEST_THIN", "1"))
def launches():
    return lib.e2eft_debug_thin_launches()
worst = 0.0
# B, H, W, Co, bias, residual, eligible
cases = [
    (2, 32, 64, 128, True, False, True),      # 16 tiles
    (1, 64, 64, 320, True, False, True),      # 16 x 3 tiles: the weights are reloaded when the N tile changes; last N tile half empty
    (3, 16, 96, 136, False, False, True),     # 18 x 2 tiles, 8 columns in the 2nd N tile, no bias, ragged per-XCD chunks
    (1, 8, 512, 128, True, False, True),      # one tile row per image
    (17, 8, 32, 128, True, False, True),      # one tile = one image, 17 tiles
    (2, 32, 48, 128, True, False, False),     # width 48: falls through
    (2, 32, 64, 128, True, True, False),      # a residual: falls through
]
for dtype in (torch.float16, torch.bfloat16):
    for (B, H, W, Co, hb, rs, elig) in cases:
        g = torch.Generator().manual_seed(B * 1000 + H * 10 + Co + W)
        x = q(torch.randn(B, 8, H, W, generator=g), dtype)
        w = q(torch.randn(Co, 8, 3, 3, generator=g) / 72 ** 0.5, dtype)
        b = q(torch.randn(Co, generator=g), dtype) if hb else None
        ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), stride=1, padding=1).float()
        rsv = q(torch.randn(ref.shape, generator=g), dtype) if rs else None
        if rsv is not None:
            ref = ref + rsv
        n0 = launches()
        out = ops.conv2d(nhwc(x, dtype, dev), pack_conv_weight(w, dtype, dev), None if b is None else b.to(dtype).to(dev), Co, 3, 3, 1, (1, 1, 1, 1),
                         residual=None if rsv is None else nhwc(rsv, dtype, dev))
        torch.cuda.synchronize()
        took = launches() - n0
        tiles = (B * H * W // 256) * ((Co + 127) // 128)
        e = rel_err(to_nchw(out), ref)
        ok = e <= TOL[dtype] and bool(torch.isfinite(out.float()).all())
        print("%%s conv 8->%%d %%s rel err %%.2e thin=%%d tiles=%%d %%s" %% (str(dtype)[6:], Co, (B, H, W), e, took, tiles, "ok" if ok else "FAIL"), flush=True)
        worst = max(worst, e / TOL[dtype])
        assert ok
        assert took == (1 if (EXPECT and elig and tiles >= 16) else 0), (took, tiles)

# ---- impulse responses: one nonzero input value, every tap, across tile borders: exact
for (py, px) in [(0, 0), (7, 31), (8, 32), (15, 63), (31, 0), (16, 33)]:
    x = torch.zeros(2, 8, 32, 64)
    x[1, 5, py, px] = 1.0
    x[0, 2, 31 - py, 63 - px] = -2.0
    w = torch.arange(9, dtype=torch.float32).reshape(1, 1, 3, 3).add(1.0).repeat(128, 8, 1, 1) / 16.0
    w = w * (1.0 + torch.arange(8).reshape(1, 8, 1, 1))
    ref = F.conv2d(x.double(), w.double(), None, padding=1).float()
    n0 = launches()
    out = ops.conv2d(nhwc(x, torch.float16, dev), pack_conv_weight(w, torch.float16, dev), None, 128, 3, 3, 1, (1, 1, 1, 1))
    torch.cuda.synchronize()
    assert torch.equal(to_nchw(out).float(), ref), (py, px)
    assert launches() - n0 == EXPECT
print("impulse responses exact", flush=True)

# ---- fused GroupNorm statistics
for (B, H, W, Co) in [(4, 32, 32, 128), (1, 64, 64, 320)]:
    g = torch.Generator().manual_seed(B + H + Co)
    x = q(torch.randn(B, 8, H, W, generator=g) * 3.0 + 1.5, torch.float16)
    w = q(torch.randn(Co, 8, 3, 3, generator=g) / 72 ** 0.5, torch.float16)
    bb = q(torch.randn(Co, generator=g) * 4.0, torch.float16)
    n0 = launches()
    y = ops.conv2d(nhwc(x, torch.float16, dev), pack_conv_weight(w, torch.float16, dev), bb.half().to(dev), Co, 3, 3, 1, (1, 1, 1, 1), gn_stats=True)
    took = launches() - n0
    yref = F.conv2d(x.double(), w.double(), bb.double(), stride=1, padding=1)
    assert rel_err(to_nchw(y), yref.float()) <= TOL[torch.float16]
    assert getattr(y, "_e2eft_gn", None) is not None, "no GroupNorm statistics emitted"
    ga, be = torch.ones(Co, device=dev).half(), torch.zeros(Co, device=dev).half()
    a = ops.groupnorm(y, ga, be, 32, 1e-5, True)
    b_ = ops.groupnorm(y.clone(), ga, be, 32, 1e-5, True)
    ref = F.silu(F.group_norm(to_nchw(y).double(), 32, eps=1e-5)).float()
    e1, e2 = rel_err(to_nchw(a), ref), rel_err(to_nchw(b_), ref)
    print("gn stats %%s: with partials %%.2e, own pass %%.2e thin=%%d" %% ((B, H, W, Co), e1, e2, took), flush=True)
    assert e1 < 2e-3 and e2 < 2e-3
    assert took == EXPECT

x = nhwc(q(torch.randn(2, 8, 32, 64), torch.float16), torch.float16, dev)
w = pack_conv_weight(q(torch.randn(128, 8, 3, 3) / 8.5, torch.float16), torch.float16, dev)
y1 = ops.conv2d(x, w, None, 128, 3, 3, 1, (1, 1, 1, 1), gn_stats=True)
y2 = ops.conv2d(x, w, None, 128, 3, 3, 1, (1, 1, 1, 1), gn_stats=True)
assert torch.equal(y1, y2) and torch.equal(y1._e2eft_gn.partial, y2._e2eft_gn.partial)
print("THIN CASES PASSED worst %%.2f of tolerance" %% worst)
''' % (HERE, HERE)


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, timeout=600)
    print(r.stdout[-6000:])
    assert r.returncode == 0 and "THIN CASES PASSED" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


def test_thin_input_kernel_on_small_shapes(dev):
    _run({"TEST_THIN": "1"})


def test_same_cases_on_igemm2(dev):
    _run({"TEST_THIN": "0"})
