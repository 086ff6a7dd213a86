#!/usr/bin/env python
"""bench.py — images/s of the single-step Marigold denoising path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the hot path (frozen VAE encode -> SD-v2 UNet @ t=999 on a zero latent -> x0 -> VAE decode ->
depth) over one batch of synthetic 768x768 images per rank (BASELINE.json configs[1]: batch 8, fp16).  Inputs are
resident in HBM when the timed region starts.  Weak scaling: every rank processes its own batch, no data-path
collective; time = max over ranks between barrier + synchronize on both sides.

Prints ONE JSON line on rank 0 with the `roofline` (dominant kernel = implicit-GEMM MFMA kernel, per-launch HIP-event
timing over the timed region) and `cpu_baseline` (the CPU oracle timed on the host cores) objects."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

# algorithmic forward work per image, GFLOP
WORK_GF = {768: dict(unet=2140.0, enc=2613.0, dec=5763.0), 576: dict(unet=1060.0, enc=1427.0, dec=3199.0),
           256: dict(unet=177.0, enc=273.0, dec=623.0), "576x768": dict(unet=1489.0, enc=1928.0, dec=4289.0)}
# the only number the reference publishes for this path (README.md:145-158): one 576x768 image, 121 ms on an RTX 4090 (other hardware: context)
REF_README_LATENCY_MS = 121.0
PEAK_TF = {"fp16": 2500.0, "bf16": 2500.0, "fp32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="images per rank per step")
    ap.add_argument("--res", default=None, help="R or HxW; default 768 (inference, configs[1]) / 576 (--train, configs[2])")
    ap.add_argument("--dtype", default=None, choices=["fp16", "bf16", "fp32"], help="default fp16 (inference, configs[1]) / bf16 (--train)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--detail", default=None, help="write a per-shape kernel table (TSV) to this path")
    ap.add_argument("--tiny", action="store_true", help="tiny configs (plumbing check only, not a valid benchmark)")
    ap.add_argument("--train", action="store_true", help="time the E2E-FT training step (BASELINE.json configs[2]) instead of inference")
    ap.add_argument("--micro-batch", type=int, default=None, help="--train: images per micro-step per rank (default: 32 in bf16 = the whole "
                    "configs[2] batch in one pass, 122 GB of the 288 GB HBM; 16 in fp32)")
    ap.add_argument("--accum", type=int, default=None, help="--train: gradient-accumulation micro-steps per optimizer step (default 32 / micro-batch)")
    ap.add_argument("--c4", action="store_true", help="--train preset = BASELINE.json configs[3]'s per-GPU share: 2 images per GPU at 768x768, one micro-step, activation "
                    "recompute on (`--gradient_checkpointing`, train_marigold_e2e_ft_depth.sh:11); with --gpus 8 that is the configuration itself.  The line carries the "
                    "exposed all-reduce time per gradient slice and the per-micro-step loss all-gather of train.py:559")
    ap.add_argument("--round4-paths", action="store_true", help="A/B: the round-4 forms of what round 5 replaced — fp32 attention as GEMM + softmax launches, 2x-upsample "
                    "convolutions fused into a 3x3 operand fetch (forward) and 3x3 dgrad + fold-back (backward)")
    ap.add_argument("--no-cross-attn-fold", action="store_true", help="A/B: the two-token cross-attention as q-projection + attention kernel + out-projection (rounds 1-4) "
                    "instead of its folded form (modules.Attention._fold)")
    ap.add_argument("--modality", default="depth", choices=["depth", "normals"])
    ap.add_argument("--no-direct-grads", action="store_true", help="--train, A/B: FlatAdamW(direct_grads=False) — gradients accumulated by autograd into a cleared flat buffer (round 3) instead of written into it by the backward kernels")
    ap.add_argument("--grad-ckpt", action="store_true", help="--train: activation recompute in the UNet blocks and the frozen decoder "
                    "(`--gradient_checkpointing` of the reference's training scripts); off by default: the whole batch fits in 288 GB")
    ap.add_argument("--graph", action="store_true", help="inference: replay one captured hipGraph per batch instead of launching ~1.4k kernels from the host "
                    "(measured: no difference at batch 8 - the launches already run back to back; it matters for latency at batch 1)")
    ap.add_argument("--no-image-encoder", action="store_true", help="--geowizard: feed a CLIP image embedding as input instead of running ViT-L/14")
    ap.add_argument("--geowizard", action="store_true", help="time GeoWizard joint depth+normals 1-step inference (BASELINE.json configs[4]: "
                    "dual-latent UNet with cross-domain attention, 2 images per GPU by default) instead of Marigold depth")
    ap.add_argument("--no-latency-leg", action="store_true", help="inference mode at N=1: skip the one-image 576x768 latency measurement (`latency_b1_576x768`)")
    ap.add_argument("--no-geowizard-leg", action="store_true", help="inference mode at N=1: skip the GeoWizard measurement (`geowizard`: configs[4]'s per-GPU share, "
                    "2 images at 768x768 with the CLIP image tower inside)")
    ap.add_argument("--no-train-leg", action="store_true", help="inference mode at N=1: skip the E2E-FT training-step measurements "
                    "that are appended to the JSON line as `train_step` (bf16 compute) and `train_step_fp32` (the reference recipe)")
    ap.add_argument("--plumbing-check", action="store_true", help="launcher / rendezvous / timing-reduction check without GPU work (gloo, CPU): "
                    "prints a line with value null; used by tests/test_distributed_cpu.py")
    ap.add_argument("--set-option", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B runs: e2eft_set_option before anything is launched (names: scripts/_options.py, e.g. patch_conv=0, fused_norm=0); recorded in the line's config")
    args = ap.parse_args()
    if args.no_cross_attn_fold:
        from diffusion_e2e_ft_amd import modules as _M
        _M.CROSS_ATTN_FOLD = False
    if args.round4_paths:
        from diffusion_e2e_ft_amd import autograd as _F, ops as _ops
        _F.FUSED_FP32_ATTENTION, _F.UPCONV_DGRAD_4X4, _ops.UPCONV_PHASES_ENABLED = False, False, False
    if args.set_option:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import _options
        left = _options.take(args.set_option)
        assert not left, "unknown --set-option %s" % left
    if args.c4:
        args.train = True
        args.res = args.res or 768
        args.micro_batch, args.accum, args.grad_ckpt = args.micro_batch or 2, args.accum or 1, True
    if args.res is None:
        args.res = 576 if args.train else 768
    hw = str(args.res).lower().split("x")
    args.res_h, args.res_w = (int(hw[0]), int(hw[0])) if len(hw) == 1 else (int(hw[0]), int(hw[1]))
    args.res = args.res_h if args.res_h == args.res_w else "%dx%d" % (args.res_h, args.res_w)
    if args.dtype is None:
        args.dtype = "bf16" if args.train else "fp16"
    return args


T_START = time.time()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launcher_command(gpus, argv, port):
    """the command `bench.py --gpus N` re-executes itself under when it was started as ONE process: one rank per GPU over RCCL, the
    launch line of the task contract (the reference: `accelerate launch --multi_gpu`, training/scripts/multi_gpu.yaml:1-15)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def maybe_spawn_ranks(args):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N ranks itself and exits with their status."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    if not args.plumbing_check:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("bench.py: --gpus %d but this node exposes %d GPU(s); refusing to report a number for fewer ranks" % (args.gpus, have))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.exit(subprocess.call(launcher_command(args.gpus, sys.argv[1:], _free_port()), env=env))


def check_world(args, world):
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE %d - refusing to print a line whose n_gpus differs from --gpus" % (args.gpus, world))


def rank_table(rank, local_rank, world, dev):
    """[{rank, local_rank, device, name}] of every rank (all-gathered), so the JSON line shows which GPUs took part"""
    me = {"rank": rank, "local_rank": local_rank, "device": str(dev),
          "name": torch.cuda.get_device_name(dev) if (dev is not None and dev.type == "cuda") else "cpu"}
    if world == 1:
        return [me]
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, me)
    return out


def plumbing_main(args):
    """no GPU work: rendezvous (gloo), barrier, max-over-ranks, one JSON line from rank 0 - the launcher path on a CPU-only host"""
    from diffusion_e2e_ft_amd import dist as D
    rank, local_rank, world = D.init_from_env(backend="gloo")
    check_world(args, world)
    D.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    ranks = rank_table(rank, local_rank, world, torch.device("cpu"))
    if rank == 0:
        print(json.dumps({"metric": "plumbing check (no GPU work)", "value": None, "unit": "images/s", "n_gpus": world, "world": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3, "plumbing_check": True, "ranks": ranks}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def train_batching(args):
    mb = args.micro_batch or (16 if args.dtype == "fp32" else 32)
    acc = args.accum or max(1, 32 // mb)
    return mb, acc


def build_pipeline(dev, dtype, tiny):
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    from diffusion_e2e_ft_amd.vae import AutoencoderKL
    from diffusion_e2e_ft_amd.scheduler import DDIMScheduler
    from diffusion_e2e_ft_amd.pipeline import MarigoldPipeline
    from diffusion_e2e_ft_amd.synth import init_synthetic_
    ucfg = dict(in_channels=8)
    vcfg = {}
    if tiny:
        ucfg.update(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128)
        vcfg.update(block_out_channels=(32, 64, 128, 128))
    with torch.device(dev):
        unet = UNet2DConditionModel(**ucfg).to(dtype)
        vae = AutoencoderKL(**vcfg).to(dtype)
    init_synthetic_(unet, seed=1234)
    init_synthetic_(vae, seed=4321)
    pipe = MarigoldPipeline(unet.eval(), vae.eval(), DDIMScheduler())
    xdim = unet.config.cross_attention_dim
    g = torch.Generator(device=dev).manual_seed(0)
    pipe.empty_text_embed = (0.5 * torch.randn((1, 2, xdim), generator=g, device=dev)).to(dtype)  # [1,2,1024] like the empty prompt
    return pipe


def cpu_baseline(res_hint, budget_s=75.0):
    """CPU oracle (oracle/ - the restatement of the reference's diffusers CPU path, pinned to the reference's wiring by
    tests/test_reference_wiring_cpu.py) on this host's cores: 1 warm-up + up to 3 timed runs per case, MEDIAN reported, in
    torch's default NCHW layout (what the reference would run) and in channels_last.  Bounded: a case stops adding runs once the whole leg
    has used `budget_s` seconds (at least one timed run each)."""
    import statistics
    from oracle import config, unet_ref, vae_ref, pipeline_ref, synth
    cores = min(os.cpu_count() or 1, 32)   # torch's CPU convs get slower, not faster, beyond a few dozen threads
    torch.set_num_threads(cores)
    t_leg = time.perf_counter()
    usd = synth.fast_state_dict(unet_ref.unet_param_shapes(config.SD2_UNET), seed=1234)   # values are irrelevant for timing
    vsd = synth.fast_state_dict(vae_ref.vae_param_shapes(config.SD_VAE), seed=4321)
    cl = lambda sd: {k: (v.contiguous(memory_format=torch.channels_last) if v.dim() == 4 else v) for k, v in sd.items()}
    sds = {"nchw": (usd, vsd), "channels_last": (cl(usd), cl(vsd))}

    def run(res, layout):
        rgb, ctx = synth.synth_inputs(1, res, res, 2, 1024, seed=0)
        if layout == "channels_last":
            rgb = rgb.contiguous(memory_format=torch.channels_last)
        u, v = sds[layout]
        t0 = time.perf_counter()
        with torch.no_grad():
            pipeline_ref.single_infer_ref(u, config.SD2_UNET, v, config.SD_VAE, rgb, ctx)
        return time.perf_counter() - t0

    def case(res, layout, warm=True):
        if warm:
            run(res, layout)
        ts = []
        for _ in range(3):
            ts.append(run(res, layout))
            if time.perf_counter() - t_leg > budget_s:
                break
        return statistics.median(ts), len(ts)

    run(64, "nchw")  # thread pool, allocator, first touch of the 3.8 GB of weights
    out = {}
    for layout in ("nchw", "channels_last"):
        out[(256, layout)] = case(256, layout)          # BASELINE.json configs[0]: one 256x256 image, fp32, CPU
    tf256 = sum(WORK_GF[256].values()) / 1e3
    tf768 = sum(WORK_GF[768].values()) / 1e3
    est768 = out[(256, "nchw")][0] * tf768 / tf256
    if est768 < 30.0:   # (the FLOP-scaled estimate is pessimistic: 768x768 runs at a better rate than 256x256 — 11 s measured against 20 s estimated)
        for layout in ("nchw", "channels_last"):
            out[(768, layout)] = case(768, layout, warm=time.perf_counter() - t_leg + 4 * est768 < budget_s)
        t768, n768 = out[(768, "nchw")]
        value, how = 1.0 / t768, "1 image 768x768, median of %d run(s)" % n768
    else:
        value, how = (tf256 / tf768) / out[(256, "nchw")][0], "256x256 time scaled to 768x768 by algorithmic FLOPs (%.2f / %.2f TFLOP)" % (tf256, tf768)
    table = {"%dx%d_%s" % (r, r, l): {"median_s": round(t, 4), "runs": n} for (r, l), (t, n) in out.items()}
    return dict(value=value, unit="images/s", cores=cores, kind="port",
                sample="CPU oracle (pure-torch fp32 restatement of the diffusers path, NCHW as the reference would run it), %s, 1 warm-up, %d threads; "
                       "BASELINE configs[0] (one 256x256 image): %.2f s median" % (how, cores, out[(256, "nchw")][0]),
                channels_last_value=(1.0 / out[(768, "channels_last")][0]) if (768, "channels_last") in out else None,
                runs=table, leg_seconds=round(time.perf_counter() - t_leg, 1))


def train_main(args):
    from diffusion_e2e_ft_amd import dist as D
    rank, local_rank, world = D.init_from_env()
    check_world(args, world)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback for the product path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks = rank_table(rank, local_rank, world, dev)
    line = run_train(args, rank, world, dev)
    if rank == 0:
        line["world"], line["ranks"], line["rccl_ranks"] = world, ranks, (world if world > 1 else 0)
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def run_train(args, rank, world, dev):
    """E2E-FT optimizer step: `accum` micro-steps of (frozen VAE encode -> UNet -> x0 -> frozen VAE decode -> loss -> backward) on
    `micro_batch` images each, then gradient all-reduce (overlapped), clip_grad_norm_ and AdamW (training/train.py:470-568).
    Returns the JSON line (rank 0) or None."""
    from diffusion_e2e_ft_amd import dist as D
    from diffusion_e2e_ft_amd import ops, training
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    from diffusion_e2e_ft_amd.vae import AutoencoderKL
    from diffusion_e2e_ft_amd.synth import init_synthetic_
    cdt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[args.dtype]
    if cdt == torch.float16:
        raise SystemExit("--train supports fp32 (the reference recipe, --mixed_precision no) or bf16 compute over fp32 master weights")
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    ucfg, vcfg = dict(in_channels=8), {}
    if args.tiny:
        ucfg.update(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128)
        vcfg.update(block_out_channels=(32, 64, 128, 128))
    with torch.device(dev):
        unet = UNet2DConditionModel(**ucfg)                    # fp32 master weights
        vae = AutoencoderKL(**vcfg).to(cdt)
    init_synthetic_(unet, seed=1234)
    init_synthetic_(vae, seed=4321)
    unet.train().set_compute_dtype(cdt)
    vae.eval().requires_grad_(False)
    if getattr(args, "grad_ckpt", False):
        unet.enable_gradient_checkpointing()
        vae.enable_gradient_checkpointing()
    opt = training.FlatAdamW(unet.parameters(), lr=3e-5, max_grad_norm=1.0, direct_grads=not getattr(args, "no_direct_grads", False))
    R = args.res_h
    mb, acc = train_batching(args)
    text = 0.5 * torch.randn((1, 77, unet.config.cross_attention_dim), generator=torch.Generator(device=dev).manual_seed(0), device=dev)
    batches = [training.synthetic_batch(mb, R, R, dev, seed=1000 * rank + i, dtype=cdt) for i in range(acc)]
    sched = training.lr_lambda_for_world(20000, 100, num_processes=world)      # train.py:356: both lengths x num_processes

    def step(i):
        # world > 1: every micro-step's loss is averaged over the ranks for logging, as train.py:559 does (one small all-gather, no host sync)
        return training.train_step(unet, vae, opt, batches, text, args.modality, lr_scale=sched(i), gather_loss=world > 1)

    for i in range(args.warmup):
        loss = step(i)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    timer = ops.KernelTimer()
    ops.TIMER = timer
    opt.profile_exchange = world > 1
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # step boundaries on the launch stream: per-step times -> median
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step(args.warmup + i)
        marks[i + 1].record()
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.TIMER = None
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    assert torch.isfinite(loss).all(), "non-finite loss"
    elapsed = D.max_over_ranks(elapsed, device=dev)
    peak_mem = torch.cuda.max_memory_allocated() / 2 ** 30
    exposed = opt.exchange_exposed_ms() if world > 1 else []
    ksum = timer.summary()
    if args.detail and rank == 0:
        write_detail(args.detail, timer, args.steps)
    if rank == 0:
        n_img = mb * acc * world
        sec = elapsed / args.steps
        ig = ksum.get("igemm", dict(launches=0, ms=1e-9, flops=0.0, bytes=0.0))
        achieved = ig["flops"] / (ig["ms"] * 1e-3) / 1e12 if ig["ms"] > 0 else 0.0
        peak = PEAK_TF[args.dtype]
        pipes = None
        if args.dtype == "fp32":
            # fp32 launches run on TWO instruction families of the one matrix pipe: v_mfma_f32_32x32x2_f32 (157.3 TF/s) and — csrc/f32split.hip, E2EFT_OPT_F32_SPLIT —
            # v_mfma_f32_32x32x16_f16 on two-term f16 splits (three f16 products per fp32 product; `flops` of those launches = the f16 products).  The roofline prices
            # both against the f16 peak of the pipe: an fp32-instruction flop occupies it 2500 / 157.3 times as long as an f16 one.
            split = dict(ms=0.0, flops=0.0, nominal=0.0, launches=0)
            native = dict(ms=0.0, flops=0.0, launches=0)
            for nm, lst in timer.rec.items():
                if nm != "igemm":
                    continue
                for r in lst:
                    is_split = "f32split" in str(r[4])
                    t = split if is_split else native
                    t["ms"] += r[0].elapsed_time(r[1])
                    t["flops"] += r[2]
                    t["launches"] += r[5]
                    if is_split:
                        t["nominal"] += r[6]
            tot_ms = split["ms"] + native["ms"]
            if tot_ms > 0:
                pipe_equiv = (split["flops"] + native["flops"] * (PEAK_TF["fp16"] / PEAK_TF["fp32"])) / (tot_ms * 1e-3) / 1e12
                pipes = {"f16_products_of_two_term_splits": {"ms_per_step": split["ms"] / args.steps, "launches_per_step": split["launches"] / args.steps,
                                                             "executed_tflops": split["flops"] / max(split["ms"], 1e-9) / 1e9,
                                                             "fp32_equivalent_tflops": split["nominal"] / max(split["ms"], 1e-9) / 1e9, "peak": PEAK_TF["fp16"]},
                         "fp32_matrix_instruction": {"ms_per_step": native["ms"] / args.steps, "launches_per_step": native["launches"] / args.steps,
                                                     "tflops": native["flops"] / max(native["ms"], 1e-9) / 1e9, "peak": PEAK_TF["fp32"]},
                         "note": "`achieved` / `peak` / `frac` of this roofline are matrix-pipe time at the f16 rate: f16 products as executed + fp32-instruction flops x (2500 / 157.3); "
                                 "E2EFT_OPT_F32_SPLIT = 0 (bench.py --set-option f32_split=0) keeps every fp32 launch on the fp32 instruction"}
                achieved, peak = pipe_equiv, PEAK_TF["fp16"]
        others = {k: dict(ms_per_step=v["ms"] / args.steps, launches_per_step=v["launches"] / args.steps, gbs=v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] else 0)
                  for k, v in ksum.items() if k != "igemm"}
        line = {
            "metric": "E2E-FT training step time (affine-invariant depth loss, UNet bwd): one optimizer step",
            "value": sec, "unit": "s/step", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec * 1e3,
            "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "median_ms_per_step": per_step[len(per_step) // 2], "min_ms_per_step": per_step[0], "max_ms_per_step": per_step[-1],
            "images_per_s": n_img / sec, "final_loss": float(loss), "peak_mem_gib": peak_mem,
            "config": {"workload": "E2E-FT training step (%s loss, UNet bwd) batch=%d/GPU (%d micro-steps x %d) at %dx%d, %s compute, fp32 master "
                                   "weights + flat AdamW, %s%s" % (args.modality, mb * acc, acc, mb, R, R,
                                                                                         args.dtype if args.dtype != "fp32" else "fp32 (fp32 tensors and accumulation; matrix products as exact two-term f16 splits on the f16 pipe "
                                                                                         "where the shape allows — error <= the fp32 instruction's against float64, tests/test_f32split_gpu.py — else v_mfma_f32_32x32x2_f32)",
                                                                                         "per-block activation recompute (UNet + decoder)" if getattr(args, "grad_ckpt", False) else "no activation recompute",
                                                                                         " [TINY CONFIG - NOT A VALID BENCHMARK]" if args.tiny else ""),
                       "images_per_step": n_img, "resolution": R, "parallelism": "dp%d (RCCL all-reduce of the flat fp32 gradient, overlapped)" % world},
            "gradient_exchange": {"slices": exposed, "note": "rank 0, mean over the timed steps: time the compute stream waited in step() for each slice's "
                                                            "all-reduce (HIP events either side of the wait) = the part of the exchange the backward did not hide"} if world > 1 else None,
            "roofline": {"bound": "mfma", "kernel": "igemm kernels (fwd, dgrad, wgrad GEMMs, attention-backward GEMMs)", "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None, "launches_per_step": ig["launches"] / args.steps,
                         "kernel_ms_per_step": ig["ms"] / args.steps, "algorithmic_gflop_per_step": ig.get("flops_nominal", ig["flops"]) / args.steps / 1e9,
                         "executed_gflop_per_step": ig["flops"] / args.steps / 1e9, **({"pipes": pipes} if pipes else {}), "other_kernels": others},
        }
        return line
    return None


def geowizard_main(args):
    from diffusion_e2e_ft_amd import dist as D
    rank, local_rank, world = D.init_from_env()
    check_world(args, world)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback for the product path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks = rank_table(rank, local_rank, world, dev)
    line = run_geowizard(args, rank, world, dev)
    if rank == 0:
        line["world"], line["ranks"], line["rccl_ranks"] = world, ranks, (world if world > 1 else 0)
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def run_geowizard(args, rank, world, dev):
    """GeoWizard joint depth + normals, one step: (CLIP ViT-L/14 image embedding ->) VAE encode -> UNet on the doubled batch (cross-domain joint
    self-attention, class embedding) -> two VAE decodes (geowizard_pipeline.py:232-248,252-344).  Returns the JSON line (rank 0) or None."""
    from diffusion_e2e_ft_amd import dist as D
    from diffusion_e2e_ft_amd import ops
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    from diffusion_e2e_ft_amd.vae import AutoencoderKL
    from diffusion_e2e_ft_amd.scheduler import DDIMScheduler
    from diffusion_e2e_ft_amd.pipeline import DepthNormalEstimationPipeline
    from diffusion_e2e_ft_amd.synth import init_synthetic_
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[args.dtype]
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    with torch.device(dev):
        unet = UNet2DConditionModel(in_channels=8, cross_attention_dim=768, class_embed_type="projection", projection_class_embeddings_input_dim=10,
                                    joint_attention=True).to(dtype)
        vae = AutoencoderKL().to(dtype)
    init_synthetic_(unet, seed=1234)
    init_synthetic_(vae, seed=4321)
    enc = None
    if not args.no_image_encoder:   # CLIP ViT-L/14 is on GeoWizard's per-image path (geowizard_pipeline.py:232-248,283-284)
        from diffusion_e2e_ft_amd.clip import CLIPVisionModelWithProjection
        with torch.device(dev):
            enc = CLIPVisionModelWithProjection().to(dtype)
        init_synthetic_(enc, seed=2468)
        enc.eval()
    pipe = DepthNormalEstimationPipeline(unet.eval(), vae.eval(), DDIMScheduler(), image_encoder=enc)
    if args.graph:
        pipe.enable_hip_graphs()
    B, R = (args.batch if args.batch != 8 else 2), args.res_h
    g = torch.Generator(device=dev).manual_seed(rank)
    rgb = (torch.randint(0, 256, (B, 3, R, R), generator=g, device=dev, dtype=torch.int32).float() / 255.0 * 2.0 - 1.0).to(dtype)
    emb = None if enc is not None else (0.5 * torch.randn((B, 1, 768), generator=g, device=dev)).to(dtype)
    for _ in range(max(args.warmup, 1) if args.graph else args.warmup):
        out = pipe.single_infer(rgb, 1, "indoor", img_embed=emb)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    # HIP events around the launches of the two MFMA families only: at two images per step the small UNet levels and the CLIP tower are launch-bound, and two event
    # records around EVERY launch (~2 us of stream time each, ~1.9k launches per step) cost ~15 % here (26.4 against 31.8 images/s)
    timer = ops.KernelTimer(only={"igemm", "attn"})
    if not args.graph:
        ops.TIMER = timer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = pipe.single_infer(rgb, 1, "indoor", img_embed=emb)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, device=dev)
    ops.TIMER = None
    if args.graph:   # per-kernel durations from the same steps run eagerly after the timed region (a graph replay cannot be bracketed per kernel)
        pipe.enable_hip_graphs(False)
        ops.TIMER = timer
        for _ in range(args.steps):
            out = pipe.single_infer(rgb, 1, "indoor", img_embed=emb)
        torch.cuda.synchronize()
        ops.TIMER = None
    assert torch.isfinite(out[0].float()).all() and torch.isfinite(out[1].float()).all()
    ksum = timer.summary()
    if rank == 0:
        ig = ksum.get("igemm", dict(launches=0, ms=1e-9, flops=0.0, bytes=0.0))
        achieved = ig["flops"] / (ig["ms"] * 1e-3) / 1e12 if ig["ms"] > 0 else 0.0
        at = ksum.get("attn", dict(launches=0, ms=0.0, flops=0.0))
        line = {"metric": "images/sec (768x768, GeoWizard joint depth+normals, 1-step dual-latent UNet fwd) full path",
                "value": B * world * args.steps / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f16" if args.dtype == "fp16" else args.dtype, "data": "synthetic",
                "config": {"workload": "GeoWizard joint depth+normals 1-step (dual-latent UNet, cross-domain joint attention, class embedding), batch=%d/GPU at "
                                       "%dx%d %s, random-init weights, %s" % (B, R, R, args.dtype, "CLIP ViT-L/14 image encoder (304M) inside the timed region" if enc is not None else "CLIP image embedding as input"),
                           "images_per_step": B * world, "resolution": R, "parallelism": "dp%d (image sharding, no collective)" % world,
                           "launch_mode": "hipGraph replay" if args.graph else "host launches"},
                "roofline": {"bound": "mfma", "kernel": "igemm5_kernel + igemm2_kernel", "achieved": achieved, "peak": PEAK_TF[args.dtype], "unit": "TFLOP/s",
                             "frac": achieved / PEAK_TF[args.dtype], "traffic": None, "launches_per_step": ig["launches"] / args.steps,
                             "kernel_ms_per_step": ig["ms"] / args.steps,
                             "other_kernels": {"attn": {"ms_per_step": at["ms"] / args.steps, "tflops": at["flops"] / (at["ms"] * 1e-3) / 1e12 if at["ms"] else 0.0}}}}
        return line
    return None


def write_detail(path, timer, steps):
    """per-shape kernel table: one row per (family, shape label, kernel symbol as rocprofv3 names it)"""
    rows = sorted(timer.by_label().items(), key=lambda kv: -kv[1]["ms"])
    with open(path, "w") as f:
        f.write("kernel\tlabel\tsymbol\tlaunches/step\tms/step\tTFLOP/s\tGB/s\talgorithmic_GFLOP/launch\n")
        for (name, label), d in rows:
            lab, sym = label if isinstance(label, tuple) else (label, "")
            f.write("%s\t%s\t%s\t%.1f\t%.3f\t%.1f\t%.1f\t%.2f\n" % (name, lab, sym, d["launches"] / steps, d["ms"] / steps,
                                                                      d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] else 0, d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] else 0,
                                                                      d["flops"] / max(d["launches"], 1) / 1e9))


def by_symbol(timer, steps):
    """{kernel symbol: launches/step, ms/step, TFLOP/s} of the igemm family: the rocprofv3 kernel-stats rows can be recomputed one by one"""
    out = {}
    for (name, label), d in timer.by_label().items():
        if name != "igemm" or not isinstance(label, tuple):
            continue
        a = out.setdefault(label[1], dict(launches=0, ms=0.0, flops=0.0))
        a["launches"] += d["launches"]
        a["ms"] += d["ms"]
        a["flops"] += d["flops"]
    return {k: dict(launches_per_step=v["launches"] / steps, ms_per_step=v["ms"] / steps, avg_us=v["ms"] * 1e3 / max(v["launches"], 1),
                    tflops=v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0.0) for k, v in sorted(out.items(), key=lambda kv: -kv[1]["ms"])}


def plain_launch_fraction(symbols, peak):
    """the family's fraction over the launches that ONLY convolve / multiply: igemm6's NORM = true symbols also apply the GroupNorm + SiLU of their
    input (work that is not counted as flops), so the family figure understates the MFMA kernels by what those launches spend on it"""
    ms = sum(v["ms_per_step"] for k, v in symbols.items() if not k.endswith(", true>") or not k.startswith("igemm6"))
    fl = sum(v["ms_per_step"] * v["tflops"] for k, v in symbols.items() if not k.endswith(", true>") or not k.startswith("igemm6"))
    nrm = sum(v["ms_per_step"] for k, v in symbols.items() if k.startswith("igemm6") and k.endswith(", true>"))
    if ms <= 0 or nrm <= 0:
        return {}
    return {"frac_launches_without_fused_groupnorm": fl / ms / peak, "ms_per_step_of_launches_with_fused_groupnorm": nrm}


def kernel_mix(symbols):
    """launches per step of each kernel of the igemm family ({'igemm6_kernel': 69.0, ...}) from by_symbol()'s rows"""
    mix = {}
    for sym, v in symbols.items():
        k = sym.split("<")[0]
        k = "conv3x3_narrow" if k.startswith("conv3x3_narrow") else k
        mix[k] = mix.get(k, 0.0) + v["launches_per_step"]
    return mix


def pmc_traffic(mix, want, prof_dir=None, build_id=None):
    """HBM bytes per launch of the dominant family from the PMC counters.  rocprofv3 --pmc cannot wrap its own process, so the counters are
    collected offline on exactly this workload (scripts/pmc_traffic.py) and committed under profiles/ together with the BUILD ID of the library
    they were collected on (`e2eft_build_id()`: hash of csrc/, the headers and the flags) and the launch counts per kernel of the family.  The
    figure is printed only when the committed profile's build id equals the id of the library THIS process runs (round 4 matched launch counts
    only and cited a round-3 profile as "the same build") and the launch counts agree; else null with the reason."""
    if build_id is None:
        from diffusion_e2e_ft_amd import _lib
        build_id = _lib.build_id()
    mine = build_id
    prof = prof_dir or os.path.join(ROOT, "profiles")
    files = sorted((f for f in os.listdir(prof) if f.endswith("_pmc_hbm_traffic.json")), reverse=True) if os.path.isdir(prof) else []
    if not want:
        return None, None, "no PMC profile for this workload", {}
    seen = []
    names = {"igemm6_kernel": "igemm6", "igemm5_kernel": "igemm5", "igemm2_kernel": "igemm2", "conv3x3_narrow": "conv3x3_narrow", "conv_thin_in_kernel": "conv_thin_in"}
    for fn in files:
        with open(os.path.join(prof, fn)) as f:
            pj = json.load(f)
        if pj.get("build_id") != mine:
            seen.append("%s: build id %s" % (fn, pj.get("build_id", "none (collected before round 5)")))
            continue
        k = pj["kernels"].get("igemm")
        if not k or k.get("launches_per_step") is None:
            seen.append("%s: no launch counts" % fn)
            continue
        theirs = {kern: (pj["kernels"].get(grp) or {}).get("launches_per_step", 0.0) for kern, grp in names.items()}
        ours = {kern: mix.get(kern, 0.0) for kern in names}
        seen.append("%s: %s" % (fn, {kk: vv for kk, vv in theirs.items() if vv}))
        if all(abs(theirs[kern] - ours[kern]) < 0.01 for kern in names) and abs(k["launches_per_step"] - sum(mix.values())) < 0.01:
            others = {g: {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "launches_per_step": v.get("launches_per_step")} for g, v in pj["kernels"].items() if g != "igemm"}
            return (k["hbm_bytes_per_launch"], "static: profiles/%s (rocprofv3 --pmc passes over this workload on build id %s = the id of the library this run loaded; launches per step %s there and here)"
                    % (fn, mine, {kk: vv for kk, vv in theirs.items() if vv}), pj["source"], others)
    same_build = [x for x in seen if "build id" not in x and "no launch counts" not in x]
    if same_build:
        return None, None, "no committed PMC profile of this build (library id %s) has this run's igemm launches per step %s (%s)" % (mine, {kk: vv for kk, vv in mix.items() if vv}, "; ".join(same_build[:3])), {}
    return None, None, "no committed PMC profile was collected on this build (library id %s; %s)" % (mine, "; ".join(seen[:4])), {}


def _build_id():
    from diffusion_e2e_ft_amd import _lib
    return _lib.build_id()


def latency_leg(pipe, dev, dtype, warm=5, iters=30):
    g = torch.Generator(device=dev).manual_seed(123)
    rgb1 = (torch.randint(0, 256, (1, 3, 576, 768), generator=g, device=dev, dtype=torch.int32).float() / 255.0 * 2.0 - 1.0).to(dtype)
    pipe.enable_hip_graphs()
    try:
        for _ in range(warm):
            o = pipe.single_infer(rgb1, 1, noise="zeros", normals=False)
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            o = pipe.single_infer(rgb1, 1, noise="zeros", normals=False)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
    finally:
        pipe.enable_hip_graphs(False)
    assert torch.isfinite(o.float()).all()
    ts.sort()
    med = ts[len(ts) // 2]
    return {"metric": "latency of one 576x768 image, full path (VAE encode + UNet + VAE decode), host wall clock around single_infer + synchronize",
            "value": med, "unit": "ms", "min_ms": ts[0], "p90_ms": ts[int(0.9 * (len(ts) - 1))], "iters": iters, "launch_mode": "hipGraph replay",
            "dtype": str(dtype).replace("torch.", ""), "algorithmic_tflop": sum(WORK_GF["576x768"].values()) / 1e3,
            "reference_readme_ms": REF_README_LATENCY_MS, "reference_hardware": "RTX 4090 (the reference's README.md:145-158; OTHER hardware: context, not a baseline)",
            "vs_reference_readme": REF_README_LATENCY_MS / med}


def _mark(timeline, what):
    """leg boundaries in seconds since process start: lets a GPU-busy trace sampled around this process be read (the CPU baseline leg
    keeps the GPU idle for about a minute BEFORE any GPU work)"""
    timeline.append({"t": round(time.time() - T_START, 2), "event": what})
    print("[bench %.1fs] %s" % (time.time() - T_START, what), file=sys.stderr, flush=True)


def main():
    args = parse()
    maybe_spawn_ranks(args)          # `--gpus N` from one process: re-executes under torch.distributed.run and exits
    if args.plumbing_check:
        return plumbing_main(args)
    if args.train:
        return train_main(args)
    if args.geowizard:
        return geowizard_main(args)
    from diffusion_e2e_ft_amd import dist as D
    from diffusion_e2e_ft_amd import ops
    rank, local_rank, world = D.init_from_env()
    check_world(args, world)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback for the product path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks = rank_table(rank, local_rank, world, dev)
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[args.dtype]
    timeline = []
    cpu_leg = None
    if world == 1 and not args.no_cpu_baseline:
        # the CPU leg runs FIRST (GPU idle), so that everything after its end marker is GPU work
        _mark(timeline, "cpu_baseline start (GPU idle)")
        try:
            cpu_leg = cpu_baseline(args.res)
        except Exception as e:  # the baseline must never take the GPU number down with it
            cpu_leg = {"value": None, "unit": "images/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        _mark(timeline, "cpu_baseline end")
    torch.set_num_threads(min(8, os.cpu_count() or 1))

    pipe = build_pipeline(dev, dtype, args.tiny)
    B, R, RH, RW = args.batch, args.res, args.res_h, args.res_w
    g = torch.Generator(device=dev).manual_seed(rank)
    img = torch.randint(0, 256, (B, 3, RH, RW), generator=g, device=dev, dtype=torch.int32)
    rgb = (img.float() / 255.0 * 2.0 - 1.0).to(dtype)  # resident in HBM before the timed region (marigold_pipeline.py:245)

    def step():
        return pipe.single_infer(rgb, 1, noise="zeros", normals=False)

    if args.graph:
        pipe.enable_hip_graphs()      # the timed steps replay one captured hipGraph per batch (captured during the first warm-up step)
    _mark(timeline, "inference warm-up start")
    for _ in range(max(args.warmup, 1) if args.graph else args.warmup):
        out = step()
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    _mark(timeline, "inference timed region start")
    # HIP events on the launch stream around every launch of the DOMINANT family (implicit GEMM) inside the timed region: that is what the
    # roofline is computed from.  The other kernels run un-instrumented here (two event records per launch cost ~2 us of stream time each:
    # 2 x 1.1k launches per step) and are timed in `args.steps` extra, fully instrumented steps after the region.
    timer = ops.KernelTimer(only={"igemm"})
    if not args.graph:
        ops.TIMER = timer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.TIMER = None
    _mark(timeline, "inference timed region end")
    assert torch.isfinite(out.float()).all(), "non-finite depth output"
    elapsed = D.max_over_ranks(elapsed, device=dev)
    if args.graph:
        # a hipGraph replay cannot be bracketed per kernel: the roofline's kernel durations then come from the same K steps run
        # eagerly right after the timed region
        pipe.enable_hip_graphs(False)
        ops.TIMER = timer
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        ops.TIMER = None
    ksum = timer.summary()
    full = ops.KernelTimer()          # every instrumented launch: the other kernel families, the per-shape table
    ops.TIMER = full
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    ops.TIMER = None
    fsum = full.summary()
    if args.detail and rank == 0:
        write_detail(args.detail, full, args.steps)

    if rank == 0:
        n_img = B * world * args.steps
        value = n_img / elapsed
        ig = ksum.get("igemm", dict(launches=0, ms=1e-9, flops=0.0, bytes=0.0))
        achieved = ig["flops"] / (ig["ms"] * 1e-3) / 1e12 if ig["ms"] > 0 else 0.0
        peak = PEAK_TF[args.dtype]
        lps = ig["launches"] / args.steps
        alg_bpl = ig["bytes"] / max(ig["launches"], 1)
        symbols = by_symbol(full, args.steps)
        traffic, traffic_source, traffic_note, pmc_other = pmc_traffic(kernel_mix(symbols), (B, R, args.dtype, args.tiny) == (8, 768, "fp16", False))
        extra = {}
        for k in ("attn", "attn512", "groupnorm"):
            if k in fsum and fsum[k]["ms"] > 0:
                kk = fsum[k]
                extra[k] = dict(launches_per_step=kk["launches"] / args.steps, ms_per_step=kk["ms"] / args.steps,
                                tflops=kk["flops"] / (kk["ms"] * 1e-3) / 1e12, gbs=kk["bytes"] / (kk["ms"] * 1e-3) / 1e9)
        if "attn" in extra:
            extra["attn"]["frac_of_mfma_peak"] = extra["attn"]["tflops"] / peak
            pa = pmc_other.get("attn_fwd")
            if pa:   # q, k, v read + out written once = the algorithmic bytes of an attention launch
                extra["attn"]["hbm_bytes_per_launch_pmc"] = pa["hbm_bytes_per_launch"]
        if "attn512" in extra:
            extra["attn512"]["frac_of_mfma_peak"] = extra["attn512"]["tflops"] / peak
        if "groupnorm" in extra:
            extra["groupnorm"]["frac_of_hbm_peak"] = extra["groupnorm"]["gbs"] / HBM_PEAK_GBS
        line = {
            "metric": "images/sec (768x768, 1-step UNet fwd) full path: VAE encode + SD-v2 UNet @t=999 + VAE decode",
            "value": value, "unit": "images/s", "n_gpus": world, "world": world, "ranks": ranks, "rccl_ranks": world if world > 1 else 0,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if args.dtype == "fp16" else args.dtype, "data": "synthetic", "build_id": _build_id(),
            "config": {"workload": "marigold-e2e-ft-depth 1-step inference, batch=%d/GPU at %dx%d %s, random-init SD-v2 UNet (866M) + SD VAE (84M)%s"
                                   % (B, RH, RW, args.dtype, " [TINY CONFIG - NOT A VALID BENCHMARK]" if args.tiny else ""),
                       "images_per_step": B * world, "resolution": R, "parallelism": "dp%d (image sharding, no collective)" % world,
                       "launch_mode": "hipGraph replay (one captured graph per batch shape)" if args.graph else "host launches",
                       **({"options": args.set_option} if args.set_option else {})},
            "roofline": {"bound": "mfma", "kernel": "igemm6_kernel (persistent, halo-patch 3x3 conv; ten launches also apply the GroupNorm + SiLU of their input) + igemm5_kernel (persistent) + igemm2_kernel + conv_in / conv_out kernels: implicit-GEMM conv/linear, all launches of the timed region",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         # `achieved` counts what the kernels MULTIPLY.  The 2x-upsample convolutions run as four 2x2 phase convolutions (4/9 of the reference
                         # formulation's multiply-adds); counted in the reference's formulation (SURVEY.md 8d's per-image GFLOP) the family does:
                         "achieved_reference_formulation": ig.get("flops_nominal", ig["flops"]) / (ig["ms"] * 1e-3) / 1e12 if ig["ms"] > 0 else 0.0,
                         "frac_reference_formulation": (ig.get("flops_nominal", ig["flops"]) / (ig["ms"] * 1e-3) / 1e12 / peak) if ig["ms"] > 0 else 0.0,
                         "traffic_over_algorithmic": (traffic / alg_bpl) if traffic else None,
                         "traffic_source": traffic_source, "traffic_note": traffic_note, "algorithmic_bytes_per_launch": alg_bpl,
                         "launches_per_step": lps, "kernel_ms_per_step": ig["ms"] / args.steps,
                         "algorithmic_gflop_per_step": ig["flops"] / args.steps / 1e9,
                         "instrumentation": "HIP events around the igemm launches only inside the timed region; `other_kernels` and `by_symbol` from %d extra, "
                                            "fully instrumented steps right after it" % args.steps,
                         "by_symbol": symbols, "other_kernels": extra, **plain_launch_fraction(symbols, peak)},
        }
        try:   # stage split ("also report UNet-only"), measured after the timed region
            st = pipe.stage_times_ms(rgb, repeats=3)
            line["stages"] = {"ms_per_step": st, "unet_only_images_per_s": B * world / (st["unet"] * 1e-3),
                              "note": "HIP events around VAE encode / UNet (+ v->x0) / VAE decode (+ head) of 3 extra steps on rank 0"}
        except Exception as e:
            line["stages"] = {"error": repr(e)}
        if R in WORK_GF and not args.tiny:
            tot = sum(WORK_GF[R].values())
            line["config"]["algorithmic_tflop_per_image"] = tot / 1e3
            line["effective_tflops"] = value * tot / 1e3
        if cpu_leg is not None:
            line["cpu_baseline"] = cpu_leg
        if world == 1 and not args.tiny and not args.no_latency_leg:
            # the one figure the reference publishes for this path (README.md:145-158): ONE 576x768 image.  Batch 1 is launch-bound on the
            # small UNet levels, so it is replayed from a captured hipGraph (bit-equal to the eager launches)
            try:
                _mark(timeline, "latency leg start")
                line["latency_b1_576x768"] = latency_leg(pipe, dev, dtype)
            except Exception as e:
                line["latency_b1_576x768"] = {"value": None, "error": repr(e)}
        if world == 1 and not args.no_geowizard_leg and not args.tiny:
            # BASELINE.json configs[4]'s per-GPU share (batch 16 on 8 GPUs = 2 images per GPU at 768x768): GeoWizard joint depth + normals with the
            # CLIP image tower inside the timed region, so that the driver's default run times it too
            try:
                _mark(timeline, "geowizard leg start")
                gargs = argparse.Namespace(**vars(args))
                gargs.batch, gargs.res, gargs.res_h, gargs.res_w, gargs.steps, gargs.warmup, gargs.graph, gargs.no_image_encoder = 2, 768, 768, 768, 10, 3, False, False
                gw = run_geowizard(gargs, 0, 1, dev)
                line["geowizard"] = {k: gw[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "steps", "warmup")}
                line["geowizard"]["workload"] = gw["config"]["workload"]
                line["geowizard"]["roofline"] = {k: gw["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "other_kernels")}
            except Exception as e:
                line["geowizard"] = {"value": None, "error": repr(e)}
        if world == 1 and not args.no_train_leg and not args.tiny:
            # second part of BASELINE.json's metric ("...; E2E-FT step time"): configs[2], batch 32 at 576x576, one GPU - in the reference
            # recipe's precision (`--mixed_precision no`, training/scripts/train_marigold_e2e_ft_depth.sh:15) AND with bf16 compute over
            # fp32 master weights
            del pipe, out, rgb, img
            # 10 timed steps after 2 warm-up steps each, mean (`value`) and median: VERDICT r4 (3 steps after 1 were too few).  `train_step_fp32_ckpt` is the
            # reference recipe as its script runs it: fp32, `--gradient_checkpointing`, micro-batches of 2 x 16 (train_marigold_e2e_ft_depth.sh:9-11,15)
            # order: the legs at the reference's precision first (fp32: `--mixed_precision "no"`), the bf16-compute extra last
            for key, tdt, tsteps, twarm, ckpt in (("train_step_fp32", "fp32", 10, 2, False), ("train_step_fp32_ckpt", "fp32", 10, 2, True), ("train_step", "bf16", 10, 2, False)):
                try:
                    torch.cuda.empty_cache()
                    torch.cuda.reset_peak_memory_stats()
                    _mark(timeline, "%s leg start" % key)
                    targs = argparse.Namespace(**vars(args))
                    targs.dtype, targs.res, targs.steps, targs.warmup, targs.detail, targs.micro_batch, targs.accum, targs.grad_ckpt = tdt, 576, tsteps, twarm, None, None, None, ckpt
                    targs.res_h = targs.res_w = 576
                    t = run_train(targs, 0, 1, dev)
                    line[key] = {k: t[k] for k in ("metric", "value", "unit", "ms_per_step", "median_ms_per_step", "min_ms_per_step", "max_ms_per_step", "images_per_s", "dtype", "steps", "warmup", "peak_mem_gib")}
                    line[key]["workload"] = t["config"]["workload"]
                    line[key]["roofline"] = {k: t["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "pipes") if k in t["roofline"]}
                except Exception as e:
                    line[key] = {"value": None, "error": repr(e)}
            _mark(timeline, "train legs end")
        line["timeline"] = timeline
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
