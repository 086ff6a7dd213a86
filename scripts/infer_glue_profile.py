"""Which host code launches the ATen / runtime-copy kernels of ONE inference step (configs[1]: B images at res^2, fp16): counters on the torch.Tensor methods that
launch kernels, keyed by the calling line inside this package, then the kernel list of torch.profiler (device time per ATen kernel).
Usage: python scripts/infer_glue_profile.py [B=8] [res=768]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffusion_e2e_ft_amd  # noqa: F401,E402
from diffusion_e2e_ft_amd.pipeline import MarigoldPipeline  # noqa: E402
from diffusion_e2e_ft_amd.scheduler import DDIMScheduler  # noqa: E402
from diffusion_e2e_ft_amd.synth import init_synthetic_  # noqa: E402
from diffusion_e2e_ft_amd.unet import UNet2DConditionModel  # noqa: E402
from diffusion_e2e_ft_amd.vae import AutoencoderKL  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 768
dev = torch.device("cuda", 0)
with torch.device(dev):
    unet = UNet2DConditionModel(in_channels=8)
    vae = AutoencoderKL()
init_synthetic_(unet, seed=1234)
init_synthetic_(vae, seed=4321)
pipe = MarigoldPipeline(unet.half().eval(), vae.half().eval(), DDIMScheduler())
pipe.empty_text_embed = (0.5 * torch.randn((1, 2, 1024), generator=torch.Generator().manual_seed(0))).to(dev, torch.float16)
rgb = (torch.rand((B, 3, R, R), generator=torch.Generator().manual_seed(1)) * 2 - 1).to(dev, torch.float16)
with torch.no_grad():
    for _ in range(3):
        pipe.single_infer(rgb, 1, False, noise="zeros", normals=False)
torch.cuda.synchronize()

from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    with torch.no_grad():
        pipe.single_infer(rgb, 1, False, noise="zeros", normals=False)
    torch.cuda.synchronize()
rows = [(e.self_device_time_total, e.count, e.key) for e in prof.key_averages() if e.self_device_time_total > 0 and "e2eft" not in e.key]
print("---- kernels of one step that are NOT libe2eft's: device us | launches | name")
for t, n, k in sorted(rows, key=lambda r: -r[0])[:25]:
    print("%9.0f %5d  %s" % (t, n, k[:150]))
print("total non-libe2eft device time %.0f us in %d launches" % (sum(r[0] for r in rows), sum(r[1] for r in rows)))

calls = collections.Counter()


def _caller():
    f = sys._getframe(2)
    chain = []
    while f is not None and len(chain) < 2:
        fn = f.f_code.co_filename
        if "diffusion" in fn and "scripts" not in fn:
            chain.append("%s:%d %s" % (os.path.basename(fn), f.f_lineno, f.f_code.co_name))
        f = f.f_back
    return " <- ".join(chain)


def _wrap(obj, name, tag=None):
    orig = getattr(obj, name)

    def w(*a, **k):
        r = orig(*a, **k)
        t = a[0] if a and isinstance(a[0], torch.Tensor) else None
        info = ""
        if t is not None and isinstance(r, torch.Tensor) and t.is_cuda:
            copied = r.data_ptr() != t.data_ptr() or name in ("copy_", "add_", "__iadd__", "zero_", "fill_", "mul_")
            if not copied:
                return r
            info = "%s->%s n=%d" % (str(t.dtype).replace("torch.", ""), str(r.dtype).replace("torch.", ""), r.numel())
        elif t is not None and not t.is_cuda:
            if not (isinstance(r, torch.Tensor) and r.is_cuda):
                return r
            info = "H2D n=%d" % r.numel()
        calls[(tag or name, _caller(), info if len(info) < 40 else "")] += 1
        return r
    setattr(obj, name, w)


for m in ("to", "float", "half", "copy_", "contiguous", "clone", "add", "__add__", "add_", "__iadd__", "sum", "zero_", "fill_", "mul", "__mul__", "mul_", "div", "__truediv__", "repeat",
          "__neg__", "__sub__", "__rsub__", "__rmul__", "mean", "clamp", "clip"):
    _wrap(torch.Tensor, m)
for m in ("cat", "zeros", "zeros_like", "empty_like", "full", "tensor", "as_tensor", "arange", "stack"):
    _wrap(torch, m, "torch." + m)
_wrap(torch.nn.functional, "pad")
with torch.no_grad():
    pipe.single_infer(rgb, 1, False, noise="zeros", normals=False)
torch.cuda.synchronize()
agg = collections.Counter()
for (name, where, info), n in calls.items():
    agg[(name, where)] += n
print("---- tensor-method calls of one step that launch a kernel: calls | method | calling line <- its caller")
for (name, where), n in sorted(agg.items(), key=lambda kv: -kv[1])[:60]:
    ex = [i for (nm, wh, i), c in calls.items() if nm == name and wh == where and i][:2]
    print("%5d  %-14s %-90s %s" % (n, name, where, "; ".join(ex)))
