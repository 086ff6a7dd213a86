"""Generate tests/golden/refwiring_golden.pt — outputs and caller traces produced BY THE REFERENCE'S OWN CODE (run here, on CPU, from the
repo root: `python tests/golden/make_refwiring_golden.py`; needs /root/reference).

What runs (all from the reference's source files where they lie, see tests/refimport.py; third-party leaf modules from tests/stubs):
  unet / vae     GeoWizard/geowizard/models/unet_2d_condition.py::UNet2DConditionModel (vendored diffusers wiring: unet_2d_blocks.py,
                 transformer_2d.py, attention.py incl. XFormersJointAttnProcessor) and an AutoencoderKL assembled from the vendored
                 DownEncoderBlock2D / UNetMidBlock2D / UpDecoderBlock2D, loaded with the seeded tiny-config state dicts of tests/golden_cases.py
  marigold       Marigold/marigold/marigold_pipeline.py::MarigoldPipeline.single_infer (:372-538): depth, normals, 2-step DDIM
  geowizard      GeoWizard/geowizard/models/geowizard_pipeline.py::DepthNormalEstimationPipeline.single_infer (:252-401)
  train          training/train.py:231-243 (encode_image / decode_image) and :470-566 (the body of `with accelerator.accumulate(unet)`)
                 executed line for line with the reference's loss modules (training/util/loss.py), depth and normals modality
Each pipeline / step runs over Spy-wrapped modules (tests/callertrace.py): the fixture holds the final outputs AND the trace of every
attribute read / call the reference's caller made, which tests/test_reference_callers_gpu.py replays on the product modules on the GPU.
"""
import importlib.util
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import config, synth  # noqa: E402
import golden_cases as gc  # noqa: E402
import refimport  # noqa: E402
from callertrace import Spy  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _import(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def build_models(ref, geo=False):
    ucfg = config.TINY_GEOWIZARD_UNET if geo else config.TINY_UNET
    unet = ref.UNet2DConditionModel(**refimport.ref_unet_kwargs(ucfg))
    unet.load_state_dict(gc.tiny_geo_sd() if geo else gc.tiny_unet_sd(), strict=True)
    if geo:
        unet.enable_xformers_memory_efficient_attention()      # installs XFormersJointAttnProcessor on attn1 (attention.py:416-421)
    vae = ref.AutoencoderKL(**refimport.ref_vae_kwargs(config.TINY_VAE))
    vae.load_state_dict(gc.tiny_vae_sd(), strict=True)
    return unet.eval(), vae.eval()


def scheduler(ref):
    # scheduler_config.json of the SD-v2 family with the spacing the reference forces (run.py:157-162)
    return ref.DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                             prediction_type="v_prediction", clip_sample=False, set_alpha_to_one=False, steps_offset=1,
                             timestep_spacing="trailing")


def marigold_cases(ref):
    out = {}
    unet, vae = build_models(ref)
    rgb, ctx = synth.synth_inputs(1, 64, 96, 2, 128, seed=3)
    for name, kw in (("depth", dict(num_inference_steps=1, noise="zeros", normals=False)),
                     ("normals", dict(num_inference_steps=1, noise="zeros", normals=True)),
                     ("depth_2step", dict(num_inference_steps=2, noise="zeros", normals=False))):
        log = []
        pipe = ref.MarigoldPipeline(Spy(unet, "unet", log), Spy(vae, "vae", log), scheduler(ref), None, None)
        pipe.empty_text_embed = ctx
        with torch.no_grad():
            res = pipe.single_infer(rgb, show_pbar=False, **kw)
        out[name] = {"out": res.clone(), "trace": log, "kwargs": kw}
        print("marigold", name, tuple(res.shape), len(log), "events")
    return out


def call_input():
    """uint8 image [3, 96, 144] with smooth structure + noise (a PIL image converts to exactly this, marigold_pipeline.py:222-227)"""
    g = torch.Generator().manual_seed(17)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 96), torch.linspace(0, 1, 144), indexing="ij")
    base = torch.stack([0.5 + 0.4 * torch.sin(5 * xx + 2 * yy), 0.3 + 0.6 * yy, 0.5 + 0.3 * torch.cos(7 * xx * yy)])
    return ((base + 0.08 * torch.randn(3, 96, 144, generator=g)).clamp(0, 1) * 255).round().to(torch.uint8)


def marigold_call_cases(ref):
    """MarigoldPipeline.__call__ (marigold_pipeline.py:158-353) from source: resize_max_res to 96 -> 64x96, normalise, one pass, min-max,
    resize back to 96x144, colourise"""
    import numpy as np
    out = {}
    unet, vae = build_models(ref)
    _, ctx = synth.synth_inputs(1, 64, 96, 2, 128, seed=3)
    for name, normals in (("depth", False), ("normals", True)):
        log = []
        pipe = ref.MarigoldPipeline(Spy(unet, "unet", log), Spy(vae, "vae", log), scheduler(ref), None, None)
        pipe.empty_text_embed = ctx
        res = pipe(call_input(), denoising_steps=1, ensemble_size=1, processing_res=96, match_input_res=True, resample_method="bilinear",
                   batch_size=0, color_map="Spectral", show_progress_bar=False, noise="zeros", normals=normals)
        arr = res.normal_np if normals else res.depth_np
        col = res.normal_colored if normals else res.depth_colored
        out[name] = {"np": torch.from_numpy(np.ascontiguousarray(arr)), "colored": torch.from_numpy(np.asarray(col).copy()),
                     "net_input": log[0]["args"]["__seq__"][0]["__tensor__"].clone()}
        print("marigold __call__", name, tuple(arr.shape), "colored", tuple(np.asarray(col).shape))
    return out


def geowizard_case(ref):
    unet, vae = build_models(ref, geo=True)
    rgb, emb = gc.geo_pipe_inputs()
    rgb, emb = rgb[:1], emb[:1]
    log = []
    pipe = ref.DepthNormalEstimationPipeline(Spy(unet, "unet", log), Spy(vae, "vae", log), scheduler(ref), None, None)
    pipe.img_embed = emb
    with torch.no_grad():
        depth, normal = pipe.single_infer(rgb, num_inference_steps=1, domain="indoor", show_pbar=False, noise="zeros")
    print("geowizard", tuple(depth.shape), tuple(normal.shape), len(log), "events")
    return {"depth": depth.clone(), "normal": normal.clone(), "trace": log}


class _Accelerator:
    """the five things training/train.py:470-566 asks of `accelerator`, single process"""
    device = torch.device("cpu")
    sync_gradients = True

    def accumulate(self, model):
        import contextlib
        return contextlib.nullcontext()

    def gather(self, t):
        return t

    def backward(self, loss):
        loss.backward()

    def clip_grad_norm_(self, params, max_norm):
        params = list(params)
        self.raw_grads = [p.grad.detach().clone() for p in params]      # what backward() left, before the in-place clipping
        self.grad_norm = torch.nn.utils.clip_grad_norm_(params, max_norm)
        return self.grad_norm


def train_cases(ref):
    loss_mod = _import(os.path.join(refimport.REF, "training/util/loss.py"), "ref_loss")
    out = {}
    for modality in ("depth", "normals"):
        unet, vae = build_models(ref)
        vae.requires_grad_(False)                    # train.py:304
        unet.train()                                 # train.py:306
        batch, text = gc.train_batch()
        log = []
        noise_scheduler = scheduler(ref)
        optimizer = torch.optim.AdamW(unet.parameters(), lr=3e-5, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)   # train.py:346-353 defaults
        before = {k: v.detach().clone() for k, v in unet.named_parameters() if k in gc.TRAIN_FULL_GRADS}
        ns = dict(torch=torch, unet=Spy(unet, "unet", log, vjp=True), vae=Spy(vae, "vae", log, vjp=True), batch=batch, accelerator=_Accelerator(),
                  weight_dtype=torch.float32, noise_scheduler=noise_scheduler,
                  args=types.SimpleNamespace(noise_type="zeros", modality=modality, train_batch_size=batch["rgb"].shape[0],
                                             gradient_accumulation_steps=1, max_grad_norm=1.0),
                  empty_encoding=text, ssi_loss=loss_mod.ScaleAndShiftInvariantLoss(), angular_loss_norm=loss_mod.AngularLoss(),
                  optimizer=optimizer, lr_scheduler=types.SimpleNamespace(step=lambda: None), train_loss=0.0,
                  alpha_prod=noise_scheduler.alphas_cumprod.to(dtype=torch.float32), pyramid_noise_like=None)
        ns["beta_prod"] = 1 - ns["alpha_prod"]                       # train.py:461-462
        ref.run_source("training/train.py", 231, 243, ns)            # encode_image / decode_image
        ref.run_source("training/train.py", 472, 566, ns)            # the step body, from "# RGB latent" to optimizer.zero_grad()
        after = {k: v.detach().clone() for k, v in unet.named_parameters() if k in gc.TRAIN_FULL_GRADS}
        grads = {k: g for (k, _), g in zip(unet.named_parameters(), ns["accelerator"].raw_grads)}
        out[modality] = {
            "loss": ns["loss"].detach().clone(), "estimate": ns["current_estimate"].detach()[:, :, ::4, ::4].clone(),
            "grad_norms": {k: float(v.norm()) for k, v in grads.items()},
            "grads": {k: gc.sample_grad(grads[k]) for k in gc.TRAIN_FULL_GRADS if k in grads},
            "clip_total_norm": float(ns["accelerator"].grad_norm),
            "param_delta": {k: gc.sample_grad(after[k] - before[k]) for k in before},   # AdamW(lr 3e-5) step after clip_grad_norm_(1.0)
            "trace": log,
        }
        print("train", modality, float(ns["loss"].detach()), len(grads), "grads,", len(log), "events, |g| =", float(ns["accelerator"].grad_norm))
    return out


def model_cases(ref):
    """the reference wiring on the inputs of tests/golden_cases.MODEL_CASES (whose stored outputs are the oracle's)"""
    out = {}
    unet, vae = build_models(ref)
    with torch.no_grad():
        for name, hw in (("unet_16x16", (16, 16)), ("unet_20x12", (20, 12))):
            x, ctx = gc.unet_inputs(hw)
            out[name] = unet(x, 999, ctx).sample
        x, ctx = gc.unet_inputs((8, 8), batch=3, ctx_len=77, seed=8)
        out["unet_ctx77"] = unet(x, torch.full((3,), 999), encoder_hidden_states=ctx, return_dict=False)[0]
        rgb, z = gc.vae_inputs()
        out["vae_moments"] = vae.quant_conv(vae.encoder(rgb))
        out["vae_dec"] = vae.decoder(vae.post_quant_conv(z))
        gunet, _ = build_models(ref, geo=True)
        x, ctx, cls = gc.geo_unet_inputs()
        out["geo_unet"] = gunet(x, 999, encoder_hidden_states=ctx, class_labels=cls).sample
    print("model cases:", {k: tuple(v.shape) for k, v in out.items()})
    return out


def main():
    torch.set_num_threads(8)
    torch.manual_seed(0)
    with refimport.reference_modules() as ref:
        out = {"meta": {"stub_diffusers": ref.uses_stub_diffusers, "torch": torch.__version__},
               "model": model_cases(ref), "marigold": marigold_cases(ref), "marigold_call": marigold_call_cases(ref), "geowizard": geowizard_case(ref), "train": train_cases(ref)}
    path = os.path.join(HERE, "refwiring_golden.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
