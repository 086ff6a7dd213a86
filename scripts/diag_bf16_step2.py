"""bf16 576^2 micro-step gradients against the fp32 GPU run under a list of option sets (one process, same inputs): is the error a property of a
kernel, of the timing, or of the rounding noise itself?"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import torch
from diffusion_e2e_ft_amd import training, _lib, ops
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
KEYS = ["conv_in.weight", "mid_block.resnets.1.conv2.weight", "up_blocks.3.resnets.2.norm2.weight", "conv_norm_out.bias", "conv_out.weight"]
DEF = {"thin_input_conv": 1, "patch_conv": 1, "fused_norm": 1, "igemm2_waves": 0, "persistent_grid": 0, "persistent": 1}


def run(dtype, opts, gn_stats=True, latent=None, vae_dtype=None, jitter=None, unet_dtype=None):
    """latent: 'fp32' = the UNet is fed the fp32 encoder's latent (the encoder's rounding noise removed), 'bf16' = an fp32 run fed the bf16 encoder's latent"""
    d = dict(DEF); d.update(opts)
    _options.take(["%s=%d" % kv for kv in d.items()])
    ops.GN_STATS_ENABLED = gn_stats
    u = copy.deepcopy(unet).train()
    v = copy.deepcopy(vae).eval().requires_grad_(False)
    ud = dtype if unet_dtype is None else unet_dtype
    vd = dtype if vae_dtype is None else vae_dtype
    if ud != torch.float32:
        u = u.set_compute_dtype(ud)
    if vd != torch.float32:
        v = v.to(vd)
    orig = training.encode_image
    if latent is not None:
        ve = copy.deepcopy(vae).eval().requires_grad_(False)
        if latent == "bf16":
            ve = ve.to(torch.bfloat16)
        def enc(_vae, rgb):
            with torch.no_grad():
                z = orig(ve, rgb.to(next(ve.parameters()).dtype)).float()
                if jitter is not None:     # relative perturbation of the latent at rounding level
                    gj = torch.Generator(device=z.device).manual_seed(jitter[1])
                    z = z * (1.0 + jitter[0] * torch.randn(z.shape, generator=gj, device=z.device))
                return z.to(rgb.dtype)
        training.encode_image = enc
    try:
        loss = training.e2e_ft_loss(u, v, batch, text, "depth")
    finally:
        training.encode_image = orig
    loss.backward()
    torch.cuda.synchronize()
    named = dict(u.named_parameters())
    return {k: named[k].grad.detach().double().cpu().flatten() for k in KEYS}


ref = run(torch.float32, {})
def rel(a, b): return ((a - b).norm() / b.norm()).item()
import statistics
# the encoder's mid-block attention (no_grad -> the fused d = 512 kernel, 16-bit P) replaced by exact fp32 attention in torch: does its precision matter?
_orig512 = ops.attention512
def exact512(q, k, v, scale, out=None):
    a = torch.softmax((q.float() @ k.float().transpose(-1, -2)) * scale, dim=-1) @ v.float()
    return a.to(q.dtype)
for label, fn in (("fused attn512 in the encoder (as shipped)", _orig512), ("exact fp32 attention in the encoder", exact512)):
    ops.attention512 = fn
    draws = []
    for seed in [None, 1, 2, 3, 4, 5]:
        jit = None if seed is None else (1e-3, seed)
        gq = run(torch.bfloat16, {}, True, latent="bf16", jitter=jit)
        draws.append(max(rel(gq[k], ref[k]) for k in KEYS))
    print("%-45s draws %s  median %.3f" % (label, ["%.3f" % d for d in draws], statistics.median(draws)), flush=True)
ops.attention512 = _orig512
