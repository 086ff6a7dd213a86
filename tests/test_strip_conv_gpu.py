"""igemm4.hip (3x3 / stride-1 convolutions with row-strip reuse of the A operand) against torch CPU: the shapes that stress what is new in
it — zero padding applied to fragments (row ends and image ends inside a 256-pixel tile, every tap), strips that start before / end after
the tensor, several images per tile, an odd number of strips, ragged N and M tiles, two-source concat with the source switch between
64-channel chunks, the epilogue options.  The kernel is opt-in (E2EFT_STRIP=1, see igemm4.hip for the measurements); E2EFT_STRIP=2 makes it take every eligible
convolution, so the cases run in a subprocess with that set (the variable is read once per process)."""
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
import torch
import torch.nn.functional as F
from diffusion_e2e_ft_amd import ops
from util import nhwc, to_nchw, pack_conv_weight, q, rel_err, TOL
dev = torch.device("cuda:0")
worst = 0.0
cases = [  # B, H, W, C1, C2, Co, rowadd, residual, alpha
    (2, 37, 53, 64, 0, 128, False, False, 1.0),
    (1, 16, 16, 128, 0, 128, True, True, 0.7),
    (3, 24, 40, 192, 0, 320, False, True, 1.0),
    (2, 20, 28, 64, 128, 192, True, False, 1.0),
    (1, 30, 30, 64, 0, 64, False, False, 1.0),
    (5, 9, 7, 128, 64, 128, False, False, 1.0),
    (1, 64, 96, 256, 0, 256, True, True, 1.0),
]
for dtype in (torch.float16, torch.bfloat16):
    for (B, H, W, C1, C2, Co, ra, rs, alpha) in cases:
        g = torch.Generator().manual_seed(B * 1000 + H * 10 + C1 + Co)
        x = q(torch.randn(B, C1, H, W, generator=g), dtype)
        x2 = q(torch.randn(B, C2, H, W, generator=g), dtype) if C2 else None
        w = q(torch.randn(Co, C1 + C2, 3, 3, generator=g) / ((C1 + C2) * 9) ** 0.5, dtype)
        b = q(torch.randn(Co, generator=g), dtype)
        xin = x if x2 is None else torch.cat([x, x2], dim=1)
        ref = F.conv2d(xin.double(), w.double(), b.double(), padding=1).float()
        rav = q(torch.randn(B, Co, generator=g), dtype) if ra else None
        rsv = q(torch.randn(ref.shape, generator=g), dtype) if rs else None
        if rav is not None:
            ref = ref + rav[:, :, None, None]
        ref = ref * alpha
        if rsv is not None:
            ref = ref + rsv
        out = ops.conv2d(nhwc(x, dtype, dev), pack_conv_weight(w, dtype, dev), b.to(dtype).to(dev), Co, 3, 3, 1, (1, 1, 1, 1),
                         x2=None if x2 is None else nhwc(x2, dtype, dev), rowadd=None if rav is None else rav.to(dtype).to(dev),
                         residual=None if rsv is None else nhwc(rsv, dtype, dev), alpha=alpha)
        e = rel_err(to_nchw(out), ref)
        ok = e <= TOL[dtype] and bool(torch.isfinite(out.float()).all())
        print("%%s %%s rel err %%.2e %%s" %% (str(dtype)[6:], (B, H, W, C1, C2, Co), e, "ok" if ok else "FAIL"))
        worst = max(worst, e / TOL[dtype])
        assert ok
# the fused GroupNorm statistics of a strip-kernel output feed the next GroupNorm exactly like igemm2's
x = q(torch.randn(2, 64, 32, 32, generator=torch.Generator().manual_seed(1)), torch.float16)
w = q(torch.randn(128, 64, 3, 3, generator=torch.Generator().manual_seed(2)) / 24.0, torch.float16)
y = ops.conv2d(nhwc(x, torch.float16, dev), pack_conv_weight(w, torch.float16, dev), None, 128, 3, 3, 1, (1, 1, 1, 1), gn_stats=True)
assert getattr(y, "_e2eft_gn", None) is not None, "no GroupNorm statistics emitted"
ga, be = torch.ones(128, device=dev).half(), torch.zeros(128, device=dev).half()
a = ops.groupnorm(y, ga, be, 32, 1e-5, True)
y2 = y.clone()
b_ = ops.groupnorm(y2, ga, be, 32, 1e-5, True)     # no statistics attached: the norm computes its own
assert rel_err(a.float(), b_.float()) < 2e-3
print("STRIP CASES PASSED worst %%.2f of tolerance" %% worst)
''' % (HERE, HERE)


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "STRIP CASES PASSED" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_strip_kernel_forced_on_small_shapes(dev):
    _run({"E2EFT_STRIP": "2"})


def test_same_cases_with_the_size_rule(dev):
    """E2EFT_STRIP=1: eligible AND large problems only (the last case); the rest run on igemm2 — both dispatches give the same numbers"""
    _run({"E2EFT_STRIP": "1"})
