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
VARIANTS = sys.argv[5].split(",") if len(sys.argv) > 5 else ["default"]      # hip side only: default, nostats, nopatch, nothin, nopersist, noflashbwd, nowgrad, vaeattn_fp32, vae_fp32, unet_fp32
GRAD_KEYS = ["conv_in.weight", "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight", "down_blocks.1.resnets.0.conv1.weight",
             "mid_block.attentions.0.proj_in.weight", "mid_block.resnets.1.conv2.weight", "up_blocks.1.resnets.0.conv_shortcut.weight",
             "up_blocks.2.attentions.1.transformer_blocks.0.attn2.to_k.weight", "up_blocks.3.resnets.2.norm2.weight", "conv_norm_out.bias", "conv_out.weight"]
JITTER = 1e-3
INSTANCE = os.environ.get("BF16_INSTANCE", "oracle")     # "oracle": the oracle's seeded network + CPU-generated sample (runs anywhere); "gpuseeded": see below


def rel(a, r):
    a, r = a.double(), r.double()
    return ((a - r).norm() / r.norm().clamp_min(1e-300)).item()


DUMP = os.environ.get("BF16_DUMP")          # path: save the decoder output (est) and the gradient w.r.t. it of the fp32 run and of every draw
DUMPED = {"draws": []}


def dump_draw(fw, bw, ref=False):
    if not DUMP:
        return
    f, b = fw["decoder.conv_out"], bw["decoder.conv_out"]
    if f.shape[-1] != f.shape[-2]:          # NHWC (hip side, channel-padded) -> NCHW
        f, b = f[..., :3].permute(0, 3, 1, 2), b[..., :3].permute(0, 3, 1, 2)
    rec = (f.detach().float().cpu().contiguous(), b.detach().float().cpu().contiguous())
    if ref:
        DUMPED["ref"] = rec
    else:
        DUMPED["draws"].append((rec[0].to(torch.bfloat16), rec[1].to(torch.bfloat16)))
    torch.save(DUMPED, DUMP)


SYS = {}       # name -> [sum over draws of (forward activation - reference with bf16-ROUNDED WEIGHTS in fp32 arithmetic), count]


def sys_add(fw, ref2_f, cut=None):
    for k, v in fw.items():
        if k in ref2_f:
            a, r = v.float(), ref2_f[k].float()
            if a.shape != r.shape:
                c = min(a.shape[-1], r.shape[-1])
                a, r = a[..., :c], r[..., :c]
            d = a - r
            if k in SYS:
                SYS[k][0] += d
                SYS[k][1] += 1
            else:
                SYS[k] = [d.clone(), 1]


def report(side, order, ref_f, ref_b, draws, ref2_f=None, draws2=None):
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
        if ref2_f is not None:
            f.write("# --- against fp32 ARITHMETIC ON THE bf16-ROUNDED WEIGHTS (the rounding of the weights is common to every bf16 implementation: what is left is the arithmetic)\n")
            f.write("# probe\tdir\tmedian rel L2 error vs that reference\tsystematic part: |mean over draws of the error field| / |reference|\n")
            for name in order:
                vals = [dr[0][name] for dr in (draws2 or []) if name in dr[0]]
                if vals and name in SYS:
                    r = ref2_f[name].float()
                    m = SYS[name][0] / SYS[name][1]
                    if m.shape != r.shape:
                        r = r[..., :m.shape[-1]]
                    f.write("#2 %s\tfwd\t%.4e\t%.4e\n" % (name, statistics.median(vals), (m.double().norm() / r.double().norm()).item()))
                vals = [dr[1][name] for dr in (draws2 or []) if name in dr[1]]
                if vals:
                    f.write("#2 %s\tbwd\t%.4e\n" % (name, statistics.median(vals)))
        worst = [max(dr[1][k] for k in dr[1] if k.startswith("param:")) for dr in draws]
        q = statistics.quantiles(worst, n=4) if len(worst) >= 4 else [float("nan")] * 3
        f.write("# worst sampled parameter-gradient error per draw: %s\n" % " ".join("%.3e" % w for w in worst))
        f.write("# distribution over %d draws: min %.3f quartiles %.3f / %.3f / %.3f max %.3f\n" % (len(worst), min(worst), q[0], q[1], q[2], max(worst)))


# =====================================================================================================================================
if SIDE == "cpu":
    torch.set_num_threads(min(32, os.cpu_count() or 1))     # (torch's CPU convolutions get SLOWER beyond a few dozen threads: 572 s instead of 90 s for the fp32 run on the GPU box's host)
    from oracle import config, pipeline_ref, unet_ref, vae_ref, synth
    from diffusion_e2e_ft_amd import training
    g = torch.Generator().manual_seed(9)
    text = 0.5 * torch.randn((1, 77, 1024), generator=g)
    if INSTANCE == "gpuseeded":      # the instance tests/test_fullsize_parity_gpu.py used through round 3: network and sample drawn by the DEVICE generator (needs the GPU box)
        from diffusion_e2e_ft_amd import unet as unet_mod, vae as vae_mod
        from diffusion_e2e_ft_amd.synth import init_synthetic_
        dev = torch.device("cuda:0")
        with torch.device(dev):
            un, va = unet_mod.UNet2DConditionModel(in_channels=8), vae_mod.AutoencoderKL()
        init_synthetic_(un, seed=1234); init_synthetic_(va, seed=4321)
        usd = {k: v.detach().float().cpu() for k, v in un.state_dict().items()}
        vsd = {k: v.detach().float().cpu() for k, v in va.state_dict().items()}
        batch = {k: v.cpu() for k, v in training.synthetic_batch(1, RES, RES, dev, seed=3).items()}
        del un, va
    else:
        usd = synth.synth_state_dict(unet_ref.unet_param_shapes(config.SD2_UNET), seed=1234)
        vsd = synth.synth_state_dict(vae_ref.vae_param_shapes(config.SD_VAE), seed=4321)
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

    def run(dt, jitter, round_weights=False):
        PROBES.clear()
        STATE["jitter"] = jitter
        rw = (lambda v: v.to(torch.bfloat16).to(dt)) if round_weights else (lambda v: v.to(dt))
        sd = {k: rw(v) for k, v in usd.items()}
        for k in GRAD_KEYS:
            sd[k] = rw(usd[k]).clone().requires_grad_(True)
        vs = {k: rw(v) for k, v in vsd.items()}
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
    dump_draw(ref_f, ref_b, ref=True)
    ref2_f, ref2_b = (run(torch.float32, None, round_weights=True) if not os.environ.get("BF16_NO_REF2") else (ref_f, ref_b))
    order = list(ORDER) + ["param:" + k for k in GRAD_KEYS]
    draws, draws2 = [], []
    for s in range(NDRAWS):
        fw, bw = run(torch.bfloat16, None if s == 0 else s)
        dump_draw(fw, bw)
        draws.append(({k: rel(fw[k], ref_f[k]) for k in fw}, {k: rel(bw[k], ref_b[k]) for k in bw}))
        draws2.append(({k: rel(fw[k], ref2_f[k]) for k in fw}, {k: rel(bw[k], ref2_b[k]) for k in bw}))
        sys_add(fw, ref2_f)
        del fw, bw
        report("cpu", order, ref_f, ref_b, draws, ref2_f, draws2)
        print("draw %d: worst sampled parameter gradient %.3e" % (s, max(v for k, v in draws[-1][1].items() if k.startswith("param:"))), flush=True)

# =====================================================================================================================================
elif SIDE == "hip":
    from diffusion_e2e_ft_amd import training, modules, unet as unet_mod, vae as vae_mod
    from diffusion_e2e_ft_amd.synth import init_synthetic_
    dev = torch.device("cuda:0")
    with torch.device(dev):
        unet = unet_mod.UNet2DConditionModel(in_channels=8)
        vae = vae_mod.AutoencoderKL()
    # the SAME network and the SAME sample as the cpu side: the oracle's seeded state dicts (per-key CPU generators) loaded into the product modules, inputs from
    # the CPU generator.  (The error of a bf16 run depends strongly on the instance — L1 loss: every prediction error that flips the sign of a residual flips
    # that pixel's gradient — so two sides on different random networks cannot be compared.)
    g = torch.Generator().manual_seed(9)
    text = 0.5 * torch.randn((1, 77, 1024), generator=g)
    if INSTANCE == "gpuseeded":
        init_synthetic_(unet, seed=1234)
        init_synthetic_(vae, seed=4321)
        batch = {k: v.cpu() for k, v in training.synthetic_batch(1, RES, RES, dev, seed=3).items()}
    else:
        from oracle import config, unet_ref, vae_ref, synth
        unet.load_state_dict(synth.synth_state_dict(unet_ref.unet_param_shapes(config.SD2_UNET), seed=1234))
        vae.load_state_dict(synth.synth_state_dict(vae_ref.vae_param_shapes(config.SD_VAE), seed=4321))
        batch = {k: v.cpu() for k, v in training.synthetic_batch(1, RES, RES, torch.device("cpu"), seed=3).items()}
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

    def run(dt, jitter, udt=None, vdt=None, round_weights=False):
        FW.clear(); BW.clear(); NAMES.clear()
        u = copy.deepcopy(unet).train()
        v = copy.deepcopy(vae).eval().requires_grad_(False)
        if round_weights:
            with torch.no_grad():
                for prm in list(u.parameters()) + list(v.parameters()):
                    prm.copy_(prm.to(torch.bfloat16).float())
        udt, vdt = udt or dt, vdt or dt
        if udt != torch.float32:
            u = u.set_compute_dtype(udt)
        if vdt != torch.float32:
            v = v.to(vdt)
        for n, m in u.named_modules():
            NAMES[id(m)] = "unet." + n
        for n, m in v.named_modules():
            NAMES[id(m)] = n

        def enc(vae_, rgb):
            z = orig_enc(vae_, rgb.to(vdt)).to(udt)
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
    dump_draw(ref_f, ref_b, ref=True)
    ref2_f, ref2_b = run(torch.float32, None, round_weights=True)
    order = list(ORDER) + ["param:" + k for k in GRAD_KEYS]
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import _options
    from diffusion_e2e_ft_amd import ops, autograd
    DEF = {"thin_input_conv": 1, "patch_conv": 1, "persistent": 1}
    _vae_train_attn = modules.VaeAttention.nhwc

    def vae_attn_fp32(self, x):
        """diagnostic: the decoder's 512-wide head under autograd computed by torch in fp32 from the bf16 q | k | v (the product runs bgemm + softmax_rows in bf16)"""
        import torch.nn.functional as TF
        from diffusion_e2e_ft_amd import autograd as F
        B, H, W, C = x.shape
        if not F.needs_grad(x, self.to_q.weight):
            return _vae_train_attn(self, x)
        n, x = self.group_norm.nhwc(x, split=True)
        n = n.view(B, H * W, C)
        bias = torch.cat([self.to_q.bias, self.to_k.bias, self.to_v.bias])
        qkv = F.linear(n, (self.to_q.weight, self.to_k.weight, self.to_v.weight), bias, owner=self, name="wqkv")
        q, k, v = (t.float()[:, None] for t in (qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]))
        a = TF.scaled_dot_product_attention(q, k, v)[:, 0].to(qkv.dtype)
        out = self.to_out[0](a, residual=x.reshape(B, H * W, C)).view(B, H, W, C)
        rec(self, out)
        return out

    for var in VARIANTS:
        opts = dict(DEF)
        ops.GN_STATS_ENABLED, autograd.FLASH_BACKWARD, ops.WGRAD_DIRECT = True, True, True
        modules.VaeAttention.nhwc = _vae_train_attn
        udt = vdt = torch.bfloat16
        if var == "nostats":
            ops.GN_STATS_ENABLED = False
        elif var == "nopatch":
            opts["patch_conv"] = 0
        elif var == "nothin":
            opts["thin_input_conv"] = 0
        elif var == "nopersist":
            opts["persistent"] = 0
        elif var == "noflashbwd":
            autograd.FLASH_BACKWARD = False
        elif var == "nowgrad":
            ops.WGRAD_DIRECT = False
        elif var == "vaeattn_fp32":
            modules.VaeAttention.nhwc = vae_attn_fp32
        elif var == "vae_fp32":
            vdt = torch.float32
        elif var == "unet_fp32":
            udt = torch.float32
        elif var != "default":
            raise SystemExit("unknown variant %s" % var)
        _options.take(["%s=%d" % kv for kv in opts.items()])
        draws, draws2 = [], []
        SYS.clear()
        for s in range(NDRAWS):
            fw, bw = run(torch.bfloat16, None if s == 0 else s, udt, vdt)
            dump_draw(fw, bw)
            draws.append(({k: relc(fw[k], ref_f[k]) for k in fw if k in ref_f}, {k: relc(bw[k], ref_b[k]) for k in bw if k in ref_b}))
            draws2.append(({k: relc(fw[k], ref2_f[k]) for k in fw if k in ref2_f}, {k: relc(bw[k], ref2_b[k]) for k in bw if k in ref2_b}))
            sys_add(fw, ref2_f)
            del fw, bw
            report("hip" if var == "default" else "hip_" + var, order, ref_f, ref_b, draws, ref2_f, draws2)
            print("%s draw %d: worst sampled parameter gradient %.3e" % (var, s, max(v for k, v in draws[-1][1].items() if k.startswith("param:"))), flush=True)
else:
    raise SystemExit(__doc__)
