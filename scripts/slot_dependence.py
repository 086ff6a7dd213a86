"""Does any kernel compute a sample differently depending on WHERE in the batch it lies?  The bf16 E2E-FT micro-step on a batch [a, b] and on its mirror
[b, a] (training/train.py:470-556; 576^2 by default): every probe of scripts/bf16_localise.py (block outputs, forward and backward) of sample a in slot 0
is compared BITWISE with sample a in slot 1 of the mirrored run.  Per-sample arithmetic that is independent of the slot gives 0 differing elements
everywhere; the first probe with a difference names the kernel whose rounding depends on the position (tile <-> image alignment, statistics merge order, ...).
usage: python scripts/slot_dependence.py [res=576] [batch=2] [dtype=bf16]"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diffusion_e2e_ft_amd import training, modules, unet as unet_mod, vae as vae_mod
from diffusion_e2e_ft_amd.synth import init_synthetic_

RES = int(sys.argv[1]) if len(sys.argv) > 1 else 576
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
DT = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}[sys.argv[3] if len(sys.argv) > 3 else "bf16"]
if "nosplit" in sys.argv[4:]:      # the fused d = 512 attention cuts the LAST round of its workgroups along the keys (attn512.hip "tail balancing"): which samples those are depends on the slot
    from diffusion_e2e_ft_amd import ops
    ops.ATTN512_SPLIT_TAIL = False
dev = torch.device("cuda:0")
with torch.device(dev):
    unet = unet_mod.UNet2DConditionModel(in_channels=8)
    vae = vae_mod.AutoencoderKL()
init_synthetic_(unet, seed=1234)
init_synthetic_(vae, seed=4321)
text = 0.5 * torch.randn((1, 77, 1024), generator=torch.Generator().manual_seed(9))
batch = training.synthetic_batch(B, RES, RES, dev, seed=3)
NAMES, FW, BW, ORDER = {}, {}, {}, []


def rec(mod, out, suffix=""):
    name = NAMES.get(id(mod))
    if name is None:
        return
    name += suffix
    FW[name] = out.detach()[[0, B - 1]].clone()          # only the two end slots are kept (B = 32: the probes of a whole batch would not fit)
    if name not in ORDER:
        ORDER.append(name)
    if out.requires_grad:
        def hook(gr, name=name):
            BW[name] = gr.detach()[[0, B - 1]].clone()
        out.register_hook(hook)


def patch(cls, suffix=""):
    orig = cls.nhwc

    def nhwc(self, *a, **k):
        out = orig(self, *a, **k)
        rec(self, out, suffix)
        return out
    cls.nhwc = nhwc


patch(modules.ResnetBlock2D); patch(modules.Transformer2DModel); patch(modules.VaeAttention)
patch(modules.Downsample2D, ".conv"); patch(modules.Upsample2D, ".conv")
for m in (unet_mod, vae_mod):
    def conv_w(conv, x, *a, _orig=m.conv_nhwc, **k):
        out = _orig(conv, x, *a, **k)
        rec(conv, out)
        return out
    m.conv_nhwc = conv_w


def run(b):
    FW.clear(); BW.clear(); NAMES.clear()
    u = copy.deepcopy(unet).train()
    v = copy.deepcopy(vae).eval().requires_grad_(False)
    if DT != torch.float32:
        u = u.set_compute_dtype(DT)
        v = v.to(DT)
    for n, m in u.named_modules():
        NAMES[id(m)] = "unet." + n
    for n, m in v.named_modules():
        NAMES[id(m)] = n
    loss = training.e2e_ft_loss(u, v, b, text, "depth")
    loss.backward()
    torch.cuda.synchronize()
    return dict(FW), dict(BW), loss.item()


fa, ba, la = run(batch)
fb, bb, lb = run({k: v.flip(0).contiguous() for k, v in batch.items()})
print("loss %.8f / mirrored %.8f" % (la, lb))
print("%-58s %-4s %s" % ("probe", "dir", "elements of sample 0 that differ between slot 0 and slot %d (of %s), max |diff| / max |value|" % (B - 1, "the sample's elements")))
first = None
for name in ORDER:
    for d, A, Bm in (("fwd", fa, fb), ("bwd", ba, bb)):
        if name not in A or name not in Bm:
            continue
        x, y = A[name][0].float(), Bm[name][1].float()
        nd = (x != y).sum().item()
        rel = ((x - y).abs().max() / x.abs().max().clamp_min(1e-30)).item()
        flag = ""
        if nd and first is None:
            first = (name, d)
            flag = "   <-- first difference"
        print("%-58s %-4s %d / %d   %.2e%s" % (name, d, nd, x.numel(), rel, flag))
print("first slot-dependent probe:", first)
