"""Which host code launches the ATen / runtime-copy kernels of a training step: torch.profiler with Python stacks over one E2E-FT step (full-size UNet + VAE,
any batch), grouped by (aten op, innermost frames inside this package; ops the autograd engine issues itself show `_engine_run_backward`).
Usage: python scripts/train_glue_profile.py [B] [res] [stacks|methods] [bf16|fp32]   (methods: counters on the torch.Tensor methods instead of the profiler)"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffusion_e2e_ft_amd  # noqa: F401,E402
from diffusion_e2e_ft_amd import training  # noqa: E402
from diffusion_e2e_ft_amd.synth import init_synthetic_  # noqa: E402
from diffusion_e2e_ft_amd.unet import UNet2DConditionModel  # noqa: E402
from diffusion_e2e_ft_amd.vae import AutoencoderKL  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
R = int(sys.argv[2]) if len(sys.argv) > 2 else 256
MODE = sys.argv[3] if len(sys.argv) > 3 else "stacks"
dev = torch.device("cuda", 0)
cdt = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "fp32") else torch.bfloat16
with torch.device(dev):
    unet = UNet2DConditionModel(in_channels=8)
    vae = AutoencoderKL().to(cdt)
init_synthetic_(unet, seed=1234)
init_synthetic_(vae, seed=4321)
unet.train().set_compute_dtype(cdt)
vae.eval().requires_grad_(False)
opt = training.FlatAdamW(unet.parameters(), lr=3e-5, max_grad_norm=1.0)
text = 0.5 * torch.randn((1, 77, 1024), generator=torch.Generator(device=dev).manual_seed(0), device=dev)
batches = [training.synthetic_batch(B, R, R, dev, seed=1, dtype=cdt)]
for i in range(2):
    training.train_step(unet, vae, opt, batches, text, "depth")
torch.cuda.synchronize()
if MODE == "stacks":
  from torch.profiler import ProfilerActivity, profile  # noqa: E402
  from torch._C._profiler import _ExperimentalConfig  # noqa: E402
  with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, experimental_config=_ExperimentalConfig(verbose=True)) as prof:
      training.train_step(unet, vae, opt, batches, text, "depth")
      torch.cuda.synchronize()
  rows = []
  for e in prof.key_averages(group_by_stack_n=14):
      if not e.key.startswith("aten::") or e.self_device_time_total <= 0:
          continue
      frames = [f for f in e.stack if ("diffusion" in f and "scripts/" not in f) or "autograd/" in f or "checkpoint" in f]
      where = " <- ".join(f.split("/")[-1][:64] for f in frames[:3]) if frames else " <- ".join(f.split("/")[-1][:50] for f in e.stack[:3])
      rows.append((e.self_device_time_total, e.count, e.key, where))
  print("device us | calls per step | op | innermost package frames (none = called from C++, e.g. the autograd engine)")
  for t, n, k, w in sorted(rows, key=lambda r: -r[1])[:45]:
      print("%9.0f %5d  %-18s %s" % (t, n, k, w))
  sys.exit(0)
# ---- who calls the tensor methods that launch ATen kernels: counters keyed on the calling line inside this package (the profiler of this build records no Python stacks)
import collections as _c  # noqa: E402
calls = _c.Counter()


def _caller():
    f = sys._getframe(2)
    while f is not None:
        fn = f.f_code.co_filename
        if "diffusion" in fn and "scripts" not in fn:
            return "%s:%d %s" % (os.path.basename(fn), f.f_lineno, f.f_code.co_name)
        f = f.f_back
    f = sys._getframe(2)
    return "%s:%d %s" % (os.path.basename(f.f_code.co_filename), f.f_lineno, f.f_code.co_name)


def _wrap(obj, name, tag=None):
    orig = getattr(obj, name)

    def w(*a, **k):
        r = orig(*a, **k)
        t = a[0] if a and isinstance(a[0], torch.Tensor) else None
        info = ""
        if t is not None and isinstance(r, torch.Tensor) and t.is_cuda:
            copied = r.data_ptr() != t.data_ptr() or name in ("copy_", "add_", "__iadd__", "zero_", "fill_")
            if not copied:
                return r
            info = "%s->%s n=%d" % (str(t.dtype).replace("torch.", ""), str(r.dtype).replace("torch.", ""), r.numel())
        elif t is not None and not t.is_cuda:
            return r
        calls[(tag or name, _caller(), info if len(info) < 40 else "")] += 1
        return r
    setattr(obj, name, w)


for m in ("to", "float", "copy_", "contiguous", "clone", "add", "__add__", "add_", "__iadd__", "sum", "flip", "zero_", "fill_", "mul", "__mul__", "div", "__truediv__", "repeat", "expand"):
    _wrap(torch.Tensor, m)
_wrap(torch, "cat")
_wrap(torch, "sum", "torch.sum")
_wrap(torch.nn.functional, "pad")
_wrap(torch, "zeros")
_wrap(torch, "zeros_like")
training.train_step(unet, vae, opt, batches, text, "depth")
torch.cuda.synchronize()
agg = _c.Counter()
for (name, where, info), n in calls.items():
    agg[(name, where)] += n
print("calls per step | method | calling line")
for (name, where), n in sorted(agg.items(), key=lambda kv: -kv[1])[:70]:
    ex = [i for (nm, wh, i), c in calls.items() if nm == name and wh == where and i][:2]
    print("%5d  %-12s %-50s %s" % (n, name, where, "; ".join(ex)))
