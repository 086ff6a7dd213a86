"""After one backward of the full-size E2E-FT micro-step: which parameters' gradients were NOT born in FlatAdamW's flat buffer (autograd.grad_sink)."""
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
dev = torch.device("cuda", 0)
cdt = torch.bfloat16
with torch.device(dev):
    unet = UNet2DConditionModel(in_channels=8)
    vae = AutoencoderKL().to(cdt)
init_synthetic_(unet, seed=1234)
init_synthetic_(vae, seed=4321)
unet.train().set_compute_dtype(cdt)
vae.eval().requires_grad_(False)
opt = training.FlatAdamW(unet.parameters(), lr=3e-5, max_grad_norm=1.0)
text = 0.5 * torch.randn((1, 77, 1024), generator=torch.Generator(device=dev).manual_seed(0), device=dev)
batch = training.synthetic_batch(B, R, R, dev, seed=1, dtype=cdt)
training.train_step(unet, vae, opt, [batch], text, "depth")
assert all(p.grad is None for p in opt.params)
training.e2e_ft_loss(unet, vae, batch, text, "depth").backward()
names = {id(p): k for k, p in unet.named_parameters()}
bad = [(names[id(p)], tuple(p.shape), None if p.grad is None else p.grad.data_ptr() - opt.flat_grad.data_ptr() - 4 * o)
       for p, o in zip(opt.params, opt.offsets) if p.grad is None or p.grad.data_ptr() != opt.flat_grad.data_ptr() + 4 * o]
print("%d of %d gradients not in their slot" % (len(bad), len(opt.params)))
for b in bad[:40]:
    print(b)
