"""Where does the bf16 rounding noise of the 576^2 E2E-FT micro-step (training/train.py:470-556) enter, and is the HIP path noisier than plain torch?

Two sides, the SAME probe points (every ResNet block / transformer / VAE attention / down- and up-sampler / conv_in / conv_out output, named by its
diffusers module path), the SAME metric (relative L2 error of the forward activation and of the gradient w.r.t. that activation against an fp32 run of
the same implementation), the SAME draws (draw 0 = plain input, draw s = latent * (1 + 1e-3 * N(0,1)) with seed s: a re-draw of every rounding):

  python scripts/bf16_localise.py cpu [ndraws] [res]    the fp32 CPU oracle against the oracle run with a bf16 state dict and bf16 activations
                                                         (plain torch bf16: what the reference would compute under `--mixed_precision bf16`)
  python scripts/bf16_localise.py hip [ndraws] [res]    the HIP path in fp32 (pinned to the oracle at 4-9e-5, tests/test_fullsize_parity_gpu.py)
                                                         against the HIP path with bf16 compute over fp32 master weights + bf16 frozen VAE

Each writes <out>_<side>.tsv: one row per probe and direction with the per-draw errors and their median; `scripts/bf16_localise_merge.py` puts the two
side by side.  Test infrastructure (imports oracle/ on the cpu side only)."""
import copy
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

SIDE = sys.argv[1]
NDRAWS = int(sys.argv[2]) if len(sys.argv) > 2 else 13
RES = int(sys.argv[3]) if len(sys.argv) > 3 else 576
OUT = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "gpurun_out", "bf16_localise")
GRAD_KEYS = ["conv_in.weight", "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight", "down_blocks.1.resnets.0.conv1.weight",
             "mid_block.attentions.0.proj_in.weight", "mid_block.resnets.1.conv2.weight", "up_blocks.1.resnets.0.conv_shortcut.weight",
             "up_blocks.2.attentions.1.transformer_blocks.0.attn2.to_k.weight", "up_blocks.3.resnets.2.norm2.weight", "conv_norm_out.bias", "conv_out.weight"]
JITTER = 1e-3


def rel(a, r):
    a, r = a.double(), r.double()
    return ((a - r).norm() / r.norm().clamp_min(1e-300)).item()


def report(side, order, ref_f, ref_b, draws):
    """draws: list of (fwd dict, bwd dict) of relative errors"""
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open("%s_%s.tsv" % (OUT, side), "w") as f:
        f.write("# %s side, %d^2, %d draws (draw 0 plain, draw s: latent jitter %g seed s); relative L2 error against the fp32 run of the same implementation\n" % (side, RES, len(draws), JITTER))
        f.write("probe\tdir\tmedian\tmin\tmax\tdraws\n")
        for name in order:
            for d, which in (("fwd", 0), ("bwd", 1)):
                vals = [dr[which][name] for dr in draws if name in dr[which]]
                if vals:
                    f.write("%s\t%s\t%.4e\t%.4e\t%.4e\t%s\n" % (name, d, statistics.median(vals), min(vals), max(vals), " ".join("%.3e" % v for v in vals)))
        worst = [max(dr[1][k] for k in dr[1] if k.startswith("param:")) for dr in draws]
        q = statistics.quantiles(worst, n=4) if len(worst) >= 4 else [float("nan")] * 3
        f.write("# worst sampled parameter-gradient error per draw: %s\n" % " ".join("%.3e" % w for w in worst))
        f.write("# distribution over %d draws: min %.3f quartiles %.3f / %.3f / %.3f max %.3f\n" % (len(worst), min(worst), q[0], q[1], q[2], max(worst)))


# =====================================================================================================================================
if SIDE == "cpu":
    torch.set_num_threads(os.cpu_count())
    from oracle import config, pipeline_ref, unet_ref, vae_ref, synth
    from diffusion_e2e_ft_amd import training
    usd = synth.synth_state_dict(unet_ref.unet_param_shapes(config.SD2_UNET), seed=1234)
    vsd = synth.synth_state_dict(vae_ref.vae_param_shapes(config.SD_VAE), seed=4321)
    g = torch.Generator().manual_seed(9)
    text = 0.5 * torch.randn((1, 77, 1024), generator=g)
    batch = {k: v.cpu() for k, v in training.synthetic_batch(1, RES, RES, torch.device("cpu"), seed=3).items()}
    PROBES, ORDER = {}, []

    def pname(name):
        return name if name.split(".")[0] in ("encoder", "decoder", "quant_conv", "post_quant_conv") else "unet." + name

    def rec(name, t):
        name = pname(name)
        if t.requires_grad:
            t.retain_grad()
        PROBES[name] = t
        if name not in ORDER:
            ORDER.append(name)

    def wrap(fn, keep=lambda n: True):
        def w(sd, name, *a, **k):
            out = fn(sd, name, *a, **k)
            if keep(name):
                rec(name, out)
            return out
        return w

    keep_conv = lambda n: ".resnets." not in n and ".attentions." not in n
    _conv_w = wrap(unet_ref._conv, keep_conv)
    _res_w = wrap(unet_ref.resnet_block)
    unet_ref._conv = vae_ref._conv = _conv_w
    unet_ref.resnet_block = vae_ref.resnet_block = _res_w
    unet_ref.transformer_2d = wrap(unet_ref.transformer_2d)
    vae_ref.vae_attention = wrap(vae_ref.vae_attention)
    _enc = pipeline_ref.encode_rgb_ref
    STATE = {"jitter": None}

    def enc(vae_sd, vae_cfg, rgb):
        z = _enc(vae_sd, vae_cfg, rgb)
        if STATE["jitter"] is not None:
            gj = torch.Generator().manual_seed(STATE["jitter"])
            z = (z.float() * (1.0 + JITTER * torch.randn(z.shape, generator=gj))).to(z.dtype)
        return z
    pipeline_ref.encode_rgb_ref = enc

    def run(dt, jitter):
        PROBES.clear()
        STATE["jitter"] = jitter
        sd = {k: v.to(dt) for k, v in usd.items()}
        for k in GRAD_KEYS:
            sd[k] = usd[k].to(dt).clone().requires_grad_(True)
        vs = {k: v.to(dt) for k, v in vsd.items()}
        b = dict(batch)
        b["rgb"] = batch["rgb"].to(dt)
        t0 = time.time()
        loss, _ = pipeline_ref.train_forward_ref(sd, config.SD2_UNET, vs, config.SD_VAE, b, text.to(dt), "depth")
        loss.float().backward()
        print("  %s run (jitter %s): %.1f s, loss %.6f" % (dt, jitter, time.time() - t0, loss.item()), flush=True)
        fw = {k: v.detach() for k, v in PROBES.items()}
        bw = {k: v.grad for k, v in PROBES.items() if v.grad is not None}
        for k in GRAD_KEYS:
            bw["param:" + k] = sd[k].grad.detach()
        PROBES.clear()
        return fw, bw

    ref_f, ref_b = run(torch.float32, None)
    order = list(ORDER) + ["param:" + k for k in GRAD_KEYS]
    draws = []
    for s in range(NDRAWS):
        fw, bw = run(torch.bfloat16, None if s == 0 else s)
        draws.append(({k: rel(fw[k], ref_f[k]) for k in fw}, {k: rel(bw[k], ref_b[k]) for k in bw}))
        del fw, bw
        report("cpu", order, ref_f, ref_b, draws)
        print("draw %d: worst sampled parameter gradient %.3e" % (s, max(v for k, v in draws[-1][1].items() if k.startswith("param:"))), flush=True)

# =====================================================================================================================================
elif SIDE == "hip":
    from diffusion_e2e_ft_amd import training, modules, unet as unet_mod, vae as vae_mod
    from diffusion_e2e_ft_amd.synth import init_synthetic_
    dev = torch.device("cuda:0")
    with torch.device(dev):
        unet = unet_mod.UNet2DConditionModel(in_channels=8)
        vae = vae_mod.AutoencoderKL()
    init_synthetic_(unet, seed=1234)
    init_synthetic_(vae, seed=4321)
    g = torch.Generator().manual_seed(9)
    text = 0.5 * torch.randn((1, 77, 1024), generator=g)
    batch = {k: v.cpu() for k, v in training.synthetic_batch(1, RES, RES, dev, seed=3).items()}
    NAMES, FW, BW, ORDER = {}, {}, {}, []

    def rec(mod, out, suffix=""):
        name = NAMES.get(id(mod))
        if name is None:
            return
        name += suffix
        FW[name] = out.detach().float().clone()
        if name not in ORDER:
            ORDER.append(name)
        if out.requires_grad:
            def hook(gr, name=name):
                BW[name] = gr.detach().float().clone()
            out.register_hook(hook)

    def patch(cls, suffix=""):
        orig = cls.nhwc

        def nhwc(self, *a, **k):
            out = orig(self, *a, **k)
            rec(self, out, suffix)
            return out
        cls.nhwc = nhwc

    patch(modules.ResnetBlock2D)
    patch(modules.Transformer2DModel)
    patch(modules.VaeAttention)
    patch(modules.Downsample2D, ".conv")
    patch(modules.Upsample2D, ".conv")
    for m in (unet_mod, vae_mod):
        def conv_w(conv, x, *a, _orig=m.conv_nhwc, **k):
            out = _orig(conv, x, *a, **k)
            rec(conv, out)
            return out
        m.conv_nhwc = conv_w
    orig_enc = training.encode_image

    def run(dt, jitter):
        FW.clear(); BW.clear(); NAMES.clear()
        u = copy.deepcopy(unet).train()
        v = copy.deepcopy(vae).eval().requires_grad_(False)
        if dt != torch.float32:
            u = u.set_compute_dtype(dt)
            v = v.to(dt)
        for n, m in u.named_modules():
            NAMES[id(m)] = "unet." + n
        for n, m in v.named_modules():
            NAMES[id(m)] = n

        def enc(vae_, rgb):
            z = orig_enc(vae_, rgb)
            if jitter is None:
                return z
            gj = torch.Generator(device=z.device).manual_seed(jitter)
            return (z.float() * (1.0 + JITTER * torch.randn(z.shape, generator=gj, device=z.device))).to(z.dtype)
        training.encode_image = enc
        try:
            loss = training.e2e_ft_loss(u, v, batch, text, "depth")
        finally:
            training.encode_image = orig_enc
        loss.backward()
        torch.cuda.synchronize()
        print("  %s run (jitter %s): loss %.6f" % (dt, jitter, loss.item()), flush=True)
        named = dict(u.named_parameters())
        fw, bw = dict(FW), dict(BW)
        for k in GRAD_KEYS:
            bw["param:" + k] = named[k].grad.detach().float().clone()
        return fw, bw

    def relc(a, r):
        c = min(a.shape[-1], r.shape[-1])        # channel padding differs with the element size (16-byte rule)
        return rel(a[..., :c], r[..., :c])

    ref_f, ref_b = run(torch.float32, None)
    order = list(ORDER) + ["param:" + k for k in GRAD_KEYS]
    draws = []
    for s in range(NDRAWS):
        fw, bw = run(torch.bfloat16, None if s == 0 else s)
        draws.append(({k: relc(fw[k], ref_f[k]) for k in fw if k in ref_f}, {k: relc(bw[k], ref_b[k]) for k in bw if k in ref_b}))
        del fw, bw
        report("hip", order, ref_f, ref_b, draws)
        print("draw %d: worst sampled parameter gradient %.3e" % (s, max(v for k, v in draws[-1][1].items() if k.startswith("param:"))), flush=True)
else:
    raise SystemExit(__doc__)
