"""Calibration for tests/test_fullsize_parity_gpu.py::test_config2_576_bf16_compute_micro_step_gradients (CPU only, not collected by pytest): what does PLAIN
torch make of the same micro-step when everything — weights, activations, gradients — is bf16?  The oracle (fp32 CPU restatement) is run with a
bf16 state dict and bf16 inputs and its sampled gradients are compared with its own fp32 run, for the plain input and two 1e-3 jitters of the image.
usage: python tests/calibrate_bf16_oracle.py [resolution=256]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
torch.set_num_threads(32)
from oracle import config, pipeline_ref, unet_ref, vae_ref, synth
from diffusion_e2e_ft_amd import training
RES = int(sys.argv[1]) if len(sys.argv) > 1 else 256
KEYS = ["conv_in.weight", "mid_block.resnets.1.conv2.weight", "up_blocks.3.resnets.2.norm2.weight", "conv_norm_out.bias", "conv_out.weight"]
usd = synth.synth_state_dict(unet_ref.unet_param_shapes(config.SD2_UNET), seed=1234)
vsd = synth.synth_state_dict(vae_ref.vae_param_shapes(config.SD_VAE), seed=4321)
g = torch.Generator().manual_seed(9)
text = 0.5 * torch.randn((1, 77, 1024), generator=g)
batch = {k: v.cpu() for k, v in training.synthetic_batch(1, RES, RES, torch.device("cpu"), seed=3).items()}
def run(dt, jitter=None):
    sd = {k: v.to(dt) for k, v in usd.items()}
    for k in KEYS:
        sd[k] = usd[k].to(dt).clone().requires_grad_(True)
    vs = {k: v.to(dt) for k, v in vsd.items()}
    b = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in batch.items()}
    if jitter is not None:
        gj = torch.Generator().manual_seed(jitter)
        b["rgb"] = (b["rgb"].float() * (1 + 1e-3 * torch.randn(b["rgb"].shape, generator=gj))).to(dt)
    t0 = time.time()
    loss, _ = pipeline_ref.train_forward_ref(sd, config.SD2_UNET, vs, config.SD_VAE, b, text.to(dt), "depth")
    loss.float().backward()
    print("  %s run: %.1f s, loss %.6f" % (dt, time.time() - t0, loss.item()), flush=True)
    return {k: sd[k].grad.detach().double().flatten() for k in KEYS}
ref = run(torch.float32)
for j in (None, 1, 2):
    gq = run(torch.bfloat16, j)
    print("torch CPU bf16 oracle at %d^2 (jitter %s): rel L2 vs fp32 %s" % (RES, j, {k.split('.')[0]: "%.3f" % (((gq[k] - ref[k]).norm() / ref[k].norm()).item()) for k in KEYS}), flush=True)
