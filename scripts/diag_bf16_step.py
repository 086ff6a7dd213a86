"""Where do two kernel configurations of the bf16 576^2 micro-step diverge?  Strided samples of every leaf module's output (forward) and of every
parameter gradient, configuration A (all kernels) against B (the options given on the command line), both against the fp32 GPU run."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import torch
from diffusion_e2e_ft_amd import training, _lib
from diffusion_e2e_ft_amd.synth import init_synthetic_
from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
from diffusion_e2e_ft_amd.vae import AutoencoderKL
import _options
dev = torch.device("cuda:0")
with torch.device(dev):
    unet = UNet2DConditionModel(in_channels=8)
    vae = AutoencoderKL()
init_synthetic_(unet, seed=1234)
init_synthetic_(vae, seed=4321)
g = torch.Generator().manual_seed(9)
text = 0.5 * torch.randn((1, 77, 1024), generator=g)
batch = {k: v.cpu() for k, v in training.synthetic_batch(1, 576, 576, dev, seed=3).items()}


def run(dtype, opts):
    for k in range(9):
        pass
    defaults = {"thin_input_conv": 1, "patch_conv": 1, "fused_norm": 1, "igemm2_waves": 0}
    defaults.update(opts)
    _options.take(["%s=%d" % kv for kv in defaults.items()])
    u = copy.deepcopy(unet).train()
    v = copy.deepcopy(vae).eval().requires_grad_(False)
    if dtype != torch.float32:
        u = u.set_compute_dtype(dtype)
        v = v.to(dtype)
    rec = []
    hooks = []
    def mk(name):
        def hook(mod, inp, out):
            t = out[0] if isinstance(out, (tuple, list)) else out
            if hasattr(t, "sample"):
                t = t.sample
            if isinstance(t, torch.Tensor) and t.is_floating_point():
                f = t.detach().float().flatten()
                step = max(1, f.numel() // 8192)
                rec.append((name, f[::step][:8192].double().cpu(), f.norm().item()))
        return hook
    for prefix, m in (("vae.", v), ("unet.", u)):
        for n, mod in m.named_modules():
            if len(list(mod.children())) == 0:
                hooks.append(mod.register_forward_hook(mk(prefix + n)))
    loss = training.e2e_ft_loss(u, v, batch, text, "depth")
    loss.backward()
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    grads = {k: p.grad.detach().double().cpu().flatten() for k, p in u.named_parameters() if p.grad is not None and (k.endswith("conv1.weight") or k.endswith("norm2.weight") or "conv_out" in k or "conv_in" in k or k.endswith("to_q.weight"))}
    return loss.item(), rec, grads


ref_loss, ref_rec, ref_g = run(torch.float32, {})
optsB = dict((a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[1:])
la, ra, ga = run(torch.bfloat16, {})
lb, rb, gb = run(torch.bfloat16, optsB)
print("loss fp32 %.6f  A %.6f  B %.6f" % (ref_loss, la, lb))
def rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()
print("%-70s %10s %10s %10s" % ("module (forward order, every 12th + jumps)", "A vs fp32", "B vs fp32", "A vs B"))
prev = 0.0
for i, ((n, sa, _), (_, sb, _), (_, sr, _)) in enumerate(zip(ra, rb, ref_rec)):
    ea, eb, eab = rel(sa, sr), rel(sb, sr), rel(sa, sb)
    if i % 12 == 0 or ea > 2.0 * prev + 1e-3 or i == len(ra) - 1:
        print("%-70s %10.3e %10.3e %10.3e" % (n[-70:], ea, eb, eab))
    prev = ea
print("gradients (every 8th):")
for i, k in enumerate(sorted(ga)):
    if i % 8 == 0:
        print("%-70s %10.3e %10.3e %10.3e" % (k[-70:], rel(ga[k], ref_g[k]), rel(gb[k], ref_g[k]), rel(ga[k], gb[k])))
