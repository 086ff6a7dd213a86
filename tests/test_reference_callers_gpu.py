"""Drop-in boundary proof on the GPU: what the REFERENCE'S caller code does with `unet` / `vae` is
replayed on the PRODUCT modules, and the results must equal what the reference's own modules returned.

tests/golden/refwiring_golden.pt was recorded by executing the reference's sources in place (tests/golden/make_refwiring_golden.py:
MarigoldPipeline.single_infer, DepthNormalEstimationPipeline.single_infer, training/train.py:470-566 with the reference's loss
modules) over Spy-wrapped modules built from the reference's vendored wiring.  Here, on the GPU box (where /root/reference does not
exist), every logged attribute read and call is made on libe2eft's modules with the logged arguments — positional / keyword forms,
0-dim int64 tensor timesteps, `return_dict=False`, `class_labels=` as the reference passes them:

  * inference traces: each call's output within 1e-3 relative (fp32) of the reference module's; the product pipelines' own
    `single_infer` against the reference pipeline's final depth / normals;
  * training trace: each call's output AND its vector-Jacobian product (the gradient autograd sent into the output -> the gradient
    it returned for the inputs) against the reference run; UNet parameter gradients after the step's backward (all 686 norms + sampled
    tensors) and the AdamW update after clip_grad_norm_(1.0) against what the reference's optimizer did;
  * the same module objects wrapped in torch DistributedDataParallel (world 1, RCCL) — `accelerator.prepare` (train.py:369-371) —
    give identical gradients.
"""
import os

import pytest
import torch

import golden_cases as gc
from callertrace import first_tensor, resolve, unpack
from oracle import config
from util import rel_err

pytestmark = pytest.mark.gpu
FX = torch.load(os.path.join(os.path.dirname(__file__), "golden", "refwiring_golden.pt"), weights_only=False)
TOL = 1e-3           # north_star: 1e-3 relative in strict fp32


def _models(dev, geo=False, train=False):
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    from diffusion_e2e_ft_amd.vae import AutoencoderKL
    unet = UNet2DConditionModel(**(config.TINY_GEOWIZARD_UNET if geo else config.TINY_UNET))
    unet.load_state_dict(gc.tiny_geo_sd() if geo else gc.tiny_unet_sd())
    vae = AutoencoderKL(**config.TINY_VAE)
    vae.load_state_dict(gc.tiny_vae_sd())
    unet, vae = unet.to(dev), vae.to(dev).eval()
    if train:
        vae.requires_grad_(False)         # train.py:304
        unet.train()                      # train.py:306
    else:
        unet.eval()
    return unet, vae


def _consumed_like_recorded(out, rec):
    """the caller took `.sample`, `[0]` or the tensor itself — the product's result must offer the same handle"""
    if isinstance(rec, dict) and "__output__" in rec:
        key = next(iter(rec["__output__"]))
        return getattr(out, key)
    if isinstance(rec, dict) and "__seq__" in rec:
        assert isinstance(out, (tuple, list))
        return out[0]
    assert isinstance(out, torch.Tensor)
    return out


def _replay(trace, roots, dev, vjp=False):
    worst = {}
    for ev in trace:
        if ev["kind"] == "getattr":
            got = resolve(roots, ev["path"])
            want = unpack(ev["value"])
            if isinstance(want, torch.Tensor):
                assert rel_err(got.float(), want) < 1e-6, ev["path"]
            else:
                assert got == want or abs(got - want) < 1e-12, (ev["path"], got, want)
        elif ev["kind"] == "getitem":
            assert resolve(roots, ev["path"])[ev["key"]] == unpack(ev["value"]), ev["path"]
        elif ev["kind"] == "method":
            assert callable(resolve(roots, ev["path"])), ev["path"]
        elif ev["kind"] == "call":
            mod = resolve(roots, ev["path"])
            need_grad = vjp and "grad_out" in ev
            args = unpack(ev["args"], dev)
            if need_grad:
                args = [a.clone().requires_grad_(True) if (isinstance(a, torch.Tensor) and i in ev.get("grad_args", {})) else a
                        for i, a in enumerate(args)]
            kwargs = {k: unpack(v, dev) for k, v in ev["kwargs"].items()}
            with torch.set_grad_enabled(need_grad):
                out = mod(*args, **kwargs)
            t = _consumed_like_recorded(out, ev["out"])
            want = first_tensor(unpack(ev["out"]))
            assert t.shape == want.shape, (ev["path"], t.shape, want.shape)
            e = rel_err(t.float(), want)
            worst[ev["path"]] = max(worst.get(ev["path"], 0.0), e)
            assert e < TOL, "%s: output rel err %.3e" % (ev["path"], e)
            if need_grad:
                t.backward(ev["grad_out"].to(dev))
                for i, g in ev["grad_args"].items():
                    eg = rel_err(args[i].grad.float(), g)
                    worst[ev["path"] + ".vjp"] = max(worst.get(ev["path"] + ".vjp", 0.0), eg)
                    assert eg < 2e-3, "%s: input-gradient rel err %.3e" % (ev["path"], eg)
        else:
            raise AssertionError(ev["kind"])
    return worst


@pytest.mark.parametrize("case", ["depth", "normals", "depth_2step"])
def test_replay_marigold_single_infer_trace(dev, case):
    """marigold_pipeline.py:372-538: vae.encoder -> vae.quant_conv -> unet(x, t, encoder_hidden_states=) .sample -> vae.post_quant_conv -> vae.decoder"""
    unet, vae = _models(dev)
    tr = FX["marigold"][case]["trace"]
    assert [e["path"] for e in tr if e["kind"] == "call"][:3] == ["vae.encoder", "vae.quant_conv", "unet"]
    print(case, _replay(tr, {"unet": unet, "vae": vae}, dev))


def test_replay_geowizard_single_infer_trace(dev):
    """geowizard_pipeline.py:252-401: unet(x, t.repeat(2), encoder_hidden_states=, class_labels=) with joint attention"""
    unet, vae = _models(dev, geo=True)
    unet.enable_xformers_memory_efficient_attention()          # what the reference calls to install the joint processor; a no-op here
    print(_replay(FX["geowizard"]["trace"], {"unet": unet, "vae": vae}, dev))


@pytest.mark.parametrize("case", ["depth", "normals", "depth_2step"])
def test_product_pipeline_equals_reference_pipeline_output(dev, case):
    from diffusion_e2e_ft_amd.pipeline import MarigoldPipeline
    from diffusion_e2e_ft_amd.scheduler import DDIMScheduler
    from oracle import synth
    unet, vae = _models(dev)
    rgb, ctx = synth.synth_inputs(1, 64, 96, 2, 128, seed=3)
    pipe = MarigoldPipeline(unet, vae, DDIMScheduler())
    pipe.empty_text_embed = ctx.to(dev)
    kw = FX["marigold"][case]["kwargs"]
    out = pipe.single_infer(rgb, kw["num_inference_steps"], False, noise=kw["noise"], normals=kw["normals"])
    want = FX["marigold"][case]["out"]
    assert out.shape == want.shape
    e = rel_err(out.float(), want)
    assert e < (2 * TOL if kw["normals"] else TOL), e


def test_product_geowizard_pipeline_equals_reference_pipeline_output(dev):
    from diffusion_e2e_ft_amd.pipeline import DepthNormalEstimationPipeline
    from diffusion_e2e_ft_amd.scheduler import DDIMScheduler
    unet, vae = _models(dev, geo=True)
    rgb, emb = gc.geo_pipe_inputs()
    pipe = DepthNormalEstimationPipeline(unet, vae, DDIMScheduler())
    pipe.img_embed = emb[:1].to(dev)
    depth, normal = pipe.single_infer(rgb[:1].to(dev), 1, "indoor", False, noise="zeros")
    assert rel_err(depth.float(), FX["geowizard"]["depth"]) < TOL
    assert rel_err(normal.float(), FX["geowizard"]["normal"]) < 2 * TOL


@pytest.mark.parametrize("modality", ["depth", "normals"])
def test_replay_train_step_trace_with_vjps(dev, modality):
    """training/train.py:470-566: outputs and vector-Jacobian products of every module call, UNet parameter gradients of the step"""
    unet, vae = _models(dev, train=True)
    fx = FX["train"][modality]
    worst = _replay(fx["trace"], {"unet": unet, "vae": vae}, dev, vjp=True)
    print(modality, worst)
    # the unet call's backward (driven by the reference run's grad_out) must have left the reference's parameter gradients
    got = {k: p.grad for k, p in unet.named_parameters()}
    assert set(got) == set(fx["grad_norms"]) and all(g is not None for g in got.values())
    bad = {k: (float(got[k].norm()), v) for k, v in fx["grad_norms"].items() if v > 1e-7 and abs(float(got[k].norm()) - v) > 2e-3 * v}
    assert not bad, list(bad.items())[:5]
    for k, g in fx["grads"].items():
        assert rel_err(gc.sample_grad(got[k].float().cpu()), g) < 2e-3, k


@pytest.mark.parametrize("modality", ["depth", "normals"])
def test_train_step_with_reference_modules_order_and_update(dev, modality):
    """the whole step on the product: e2e_ft_loss -> backward -> clip_grad_norm_(1.0) -> AdamW(3e-5), against the loss, the clipping norm
    and the parameter update the reference's step body produced (train.py:556-566)"""
    from diffusion_e2e_ft_amd import training
    unet, vae = _models(dev, train=True)
    fx = FX["train"][modality]
    batch, text = gc.train_batch()
    opt = training.FlatAdamW(unet.parameters(), lr=3e-5, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=1.0)
    before = {k: p.detach().clone() for k, p in unet.named_parameters() if k in fx["param_delta"]}
    loss = training.e2e_ft_loss(unet, vae, batch, text, modality)
    loss.backward()
    total_norm = opt.grad_norm()
    opt.step()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(fx["loss"])) < 1e-4 * abs(float(fx["loss"]))
    assert abs(total_norm - fx["clip_total_norm"]) < 2e-3 * fx["clip_total_norm"]
    for k, d in fx["param_delta"].items():
        got = gc.sample_grad((dict(unet.named_parameters())[k].detach() - before[k]).float().cpu())
        # Adam's first step moves every weight by ~lr * sign(g): compare the updates where the gradient is not at round-off level
        big = d.abs() > 0.5 * 3e-5
        if big.any():
            assert (got[big] - d[big]).abs().max() < 0.05 * 3e-5, k


def test_ddp_wrap_gives_identical_gradients(dev):
    """`accelerator.prepare(unet, ...)` wraps the UNet in DistributedDataParallel (train.py:369-371): autograd hooks on the leaf
    Parameters, forward through the wrapper, RCCL all-reduce (world 1) — gradients must equal the bare module's"""
    import torch.distributed as dist
    from diffusion_e2e_ft_amd import training
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    batch, text = gc.train_batch()
    unet, vae = _models(dev, train=True)
    training.e2e_ft_loss(unet, vae, batch, text, "depth").backward()
    want = {k: p.grad.detach().clone() for k, p in unet.named_parameters()}
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        unet2, _ = _models(dev, train=True)
        ddp = torch.nn.parallel.DistributedDataParallel(unet2, device_ids=[dev.index or 0])
        assert ddp.module is unet2 and ddp.module.conv_in.weight.shape[1] == 8
        training.e2e_ft_loss(ddp, vae, batch, text, "depth").backward()
        torch.cuda.synchronize()
        for k, p in unet2.named_parameters():
            assert torch.equal(p.grad, want[k]), k
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("normals", [False, True])
def test_call_equals_reference_call_from_source(dev, normals):
    """`MarigoldPipeline.__call__` (marigold_pipeline.py:158-353) value for value: resize_max_res (uint8 in, rounded like torchvision),
    normalisation, one pass, min-max / re-normalisation, resize back, clip, colourised image — against the reference's `__call__` run
    from source (tests/golden/make_refwiring_golden.py::marigold_call_cases)"""
    import numpy as np
    from diffusion_e2e_ft_amd.pipeline import MarigoldPipeline, resize_max_res
    from diffusion_e2e_ft_amd.scheduler import DDIMScheduler
    from oracle import synth
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_refwiring_golden import call_input
    fx = FX["marigold_call"]["normals" if normals else "depth"]
    unet, vae = _models(dev)
    _, ctx = synth.synth_inputs(1, 64, 96, 2, 128, seed=3)
    pipe = MarigoldPipeline(unet, vae, DDIMScheduler())
    pipe.empty_text_embed = ctx.to(dev)
    img = call_input()
    # what reaches the network: the reference's resized + normalised image (first vae.encoder call of its trace)
    mine = resize_max_res(img, 96, "bilinear")
    assert mine.dtype == torch.uint8 and tuple(mine.shape) == (3, 64, 96)
    net_in = mine / 255.0 * 2.0 - 1.0
    assert (net_in - fx["net_input"][0]).abs().max().item() <= 2.0 / 255.0 + 1e-6          # at most one uint8 rounding step apart
    assert ((net_in - fx["net_input"][0]).abs() > 1e-6).float().mean().item() < 0.01
    res = pipe(img, denoising_steps=1, ensemble_size=1, processing_res=96, match_input_res=True, resample_method="bilinear", batch_size=0,
               color_map="Spectral", show_progress_bar=False, noise="zeros", normals=normals)
    arr = res.normal_np if normals else res.depth_np
    col = np.asarray(res.normal_colored if normals else res.depth_colored)
    want, want_col = fx["np"].numpy(), fx["colored"].numpy()
    assert arr.shape == want.shape and arr.dtype == want.dtype and col.shape == want_col.shape and col.dtype == np.uint8
    assert (res.depth_np is None) == normals and (res.normal_colored is None) == (not normals) and res.uncertainty is None
    err = np.abs(arr - want)
    print("__call__ %s: max abs err %.2e mean %.2e" % ("normals" if normals else "depth", err.max(), err.mean()))
    assert err.max() <= (2e-2 if normals else 5e-3) and err.mean() <= 1e-3       # a rounding step of the uint8 input moves the output slightly
    assert (np.abs(col.astype(np.int32) - want_col.astype(np.int32)) > 2).mean() < 0.01


def test_find_batch_size_rule(dev):
    """util/batchsize.py:59-81: never more than the ensemble, a batch between half and all of it is cut to half"""
    from diffusion_e2e_ft_amd.pipeline import find_batch_size
    assert find_batch_size(1, 768, torch.float16) == 1 and find_batch_size(10, 768, torch.float16) == 10
    assert find_batch_size(100, 768, torch.float16) == 50 and find_batch_size(128, 768, torch.float16) == 64   # 64 > ceil(100 / 2): cut to half
    assert find_batch_size(70, 768, torch.float16) == 35            # 64 > ceil(70 / 2): two balanced passes
    assert find_batch_size(10, 2048, torch.float32) == 4
