"""E2E-FT micro-step on the GPU (training.e2e_ft_loss -> backward -> FlatAdamW) against the golden loss / UNet gradients of the
CPU oracle under torch autograd (tests/golden/train_golden.pt from tests/golden/make_golden.py): same seeded weights, batch
and text embedding.  fp32 bar: loss 1e-4, every parameter's gradient norm within 2e-3 relative, sampled gradient tensors
within 2e-3 of their largest entry; bf16 activations with fp32 master weights are checked at a stated looser tolerance."""
import gc as _pygc          # the cycle collector; the module-level name `gc` below is golden_cases
import math
import os

import pytest
import torch

import golden_cases as gc
from oracle import config
from util import rel_err

pytestmark = pytest.mark.gpu

GOLD = torch.load(os.path.join(os.path.dirname(__file__), "golden", "train_golden.pt"))


def _models(dev, unet_dtype=torch.float32):
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    from diffusion_e2e_ft_amd.vae import AutoencoderKL
    unet = UNet2DConditionModel(**config.TINY_UNET)
    unet.load_state_dict(gc.tiny_unet_sd())
    unet = unet.to(device=dev, dtype=unet_dtype).train()
    vae = AutoencoderKL(**config.TINY_VAE)
    vae.load_state_dict(gc.tiny_vae_sd())
    vae = vae.to(device=dev, dtype=unet_dtype).eval()
    vae.requires_grad_(False)          # train.py:304
    return unet, vae


def _check_grads(named, gold, tol=2e-3):
    bad = []
    floor = 1e-6 * max(gold["grad_norms"].values())
    for k, n_ref in gold["grad_norms"].items():
        g = named[k].grad
        assert g is not None, k
        n = g.float().norm().item()
        if abs(n - n_ref) > tol * n_ref + floor:
            bad.append((k, n, n_ref))
    assert not bad, bad[:10]
    for k, ref in gold["grads"].items():
        if gold["grad_norms"][k] <= floor:      # rounding noise in the oracle (e.g. softmax over a single key: exactly 0 here)
            continue
        e = rel_err(gc.sample_grad(named[k].grad.float()), ref)
        assert e < tol, (k, e)


@pytest.mark.parametrize("modality", ["depth", "normals"])
def test_micro_step_gradients_fp32(dev, modality):
    from diffusion_e2e_ft_amd import training
    unet, vae = _models(dev)
    batch, text = gc.train_batch()
    gold = GOLD[modality]
    loss, est = training.e2e_ft_loss(unet, vae, batch, text, modality, return_estimate=True)
    assert abs(loss.item() - gold["loss"].item()) <= 1e-4 * abs(gold["loss"].item()), (loss.item(), gold["loss"].item())
    assert rel_err(est[:, :, ::4, ::4].float(), gold["estimate"]) < 1e-3
    loss.backward()
    _check_grads(dict(unet.named_parameters()), gold)
    for p in vae.parameters():
        assert p.grad is None


def test_geowizard_micro_step_gradients_fp32(dev):
    """doubled batch, cross-domain joint self-attention, projection class embedding, 0.5 * SSI + angular on inverted normals
    (GeoWizard/geowizard/training/train_depth_normal.py:597-768)"""
    from diffusion_e2e_ft_amd import training
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel
    unet = UNet2DConditionModel(**config.TINY_GEOWIZARD_UNET)
    unet.load_state_dict(gc.tiny_geo_sd())
    unet = unet.to(dev).train()
    _, vae = _models(dev)
    batch, emb = gc.geo_train_inputs()
    gold = GOLD["geowizard"]
    loss, ssi, ang = training.geowizard_e2e_ft_loss(unet, vae, batch, emb, "indoor", return_parts=True)
    assert abs(ssi.item() - gold["ssi"].item()) <= 1e-4 * abs(gold["ssi"].item())
    assert abs(ang.item() - gold["angular"].item()) <= 1e-4 * abs(gold["angular"].item())
    assert abs(loss.item() - gold["loss"].item()) <= 1e-4 * abs(gold["loss"].item())
    loss.backward()
    _check_grads(dict(unet.named_parameters()), gold)


def test_micro_step_bf16_compute_over_fp32_master_weights(dev):
    """set_compute_dtype(bf16): fp32 parameters, bf16 activations and kernels; gradients arrive in fp32 with the parameter shapes"""
    from diffusion_e2e_ft_amd import training
    unet, vae = _models(dev)
    unet.set_compute_dtype(torch.bfloat16)
    vae = vae.to(torch.bfloat16)
    batch, text = gc.train_batch()
    gold = GOLD["depth"]
    loss = training.e2e_ft_loss(unet, vae, batch, text, "depth")
    assert abs(loss.item() - gold["loss"].item()) <= 5e-2 * abs(gold["loss"].item())
    loss.backward()
    cosines = {}
    for k, p in unet.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32 and p.grad.shape == p.shape, k
    named = dict(unet.named_parameters())
    for k, ref in gold["grads"].items():
        g = gc.sample_grad(named[k].grad).cpu()
        cosines[k] = torch.nn.functional.cosine_similarity(g.flatten(), ref.flatten(), dim=0).item()
    assert min(cosines.values()) > 0.75 and sum(cosines.values()) / len(cosines) > 0.88, cosines


def test_micro_step_bf16_activations(dev):
    """bf16 model: gradients agree with the fp32 golden in direction (cosine > 0.75 per sampled tensor, > 0.88 on average; everything incl. the weights is bf16 here)"""
    from diffusion_e2e_ft_amd import training
    unet, vae = _models(dev, torch.bfloat16)
    batch, text = gc.train_batch()
    gold = GOLD["depth"]
    loss = training.e2e_ft_loss(unet, vae, batch, text, "depth")
    assert abs(loss.item() - gold["loss"].item()) <= 5e-2 * abs(gold["loss"].item())
    loss.backward()
    named = dict(unet.named_parameters())
    cosines = {}
    for k, ref in gold["grads"].items():
        g = gc.sample_grad(named[k].grad.float()).cpu()
        cosines[k] = torch.nn.functional.cosine_similarity(g.flatten(), ref.flatten(), dim=0).item()
    assert min(cosines.values()) > 0.75 and sum(cosines.values()) / len(cosines) > 0.88, cosines


def test_optimizer_step_matches_torch_adamw(dev):
    """one full train_step with FlatAdamW == the same gradients through clip_grad_norm_ + torch.optim.AdamW (train.py:561-566)"""
    from diffusion_e2e_ft_amd import training
    unet, vae = _models(dev)
    batch, text = gc.train_batch()
    before = {k: v.detach().clone() for k, v in unet.named_parameters()}
    opt = training.FlatAdamW(unet.parameters(), lr=1e-3, max_grad_norm=1.0)
    sd = unet.state_dict()
    assert all(torch.equal(sd[k], before[k]) for k in before)           # flattening keeps values and names
    loss = training.e2e_ft_loss(unet, vae, batch, text, "depth")
    loss.backward()
    grads = {k: v.grad.detach().clone() for k, v in unet.named_parameters()}
    gn = opt.grad_norm()
    assert abs(gn - math.sqrt(sum(float(g.double().pow(2).sum()) for g in grads.values()))) < 1e-4 * gn
    opt.step()
    opt.zero_grad()
    ref_params = [torch.nn.Parameter(before[k].clone()) for k in before]
    for p, k in zip(ref_params, before):
        p.grad = grads[k].clone()
    torch.nn.utils.clip_grad_norm_(ref_params, 1.0)
    ropt = torch.optim.AdamW(ref_params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    ropt.step()
    for p, (k, v) in zip(ref_params, unet.named_parameters()):
        assert rel_err(v, p) < 1e-5, k
        assert v.grad is None                      # FlatAdamW(direct_grads=True).zero_grad(): the next backward's kernels write the slots
    # the step changed the weights the kernels see: a second forward uses re-packed weights
    loss2 = training.e2e_ft_loss(unet, vae, batch, text, "depth")
    assert loss2.item() != loss.item() and math.isfinite(loss2.item())


def test_gradient_accumulation_equals_big_batch(dev):
    from diffusion_e2e_ft_amd import training
    unet, vae = _models(dev)
    batch, text = gc.train_batch()
    halves = [{k: v[i:i + 1] for k, v in batch.items()} for i in range(2)]
    # SSI / L1 mean over valid pixels of the whole batch != mean of per-image means in general, so compare like with like
    l = [training.e2e_ft_loss(unet, vae, h, text, "depth") for h in halves]
    ((l[0] + l[1]) / 2).backward()
    ref = {k: v.grad.detach().clone() for k, v in unet.named_parameters()}
    unet.zero_grad(set_to_none=True)
    opt = training.FlatAdamW(unet.parameters(), lr=0.0, max_grad_norm=0.0)
    for i, h in enumerate(halves):
        (training.e2e_ft_loss(unet, vae, h, text, "depth") / 2).backward()
    for k, v in unet.named_parameters():
        assert rel_err(v.grad, ref[k]) < 1e-5 or ref[k].abs().max() < 1e-12, k


def test_reference_style_step_with_torch_glue(dev):
    """The step body as training/train.py:470-566 writes it — torch arithmetic between `unet(...)` and `vae.decoder(...)`,
    `.mean(dim=1, keepdim=True)`, `torch.clamp`, the loss as a torch module (the oracle's restatement of loss.py, pinned to the
    reference), `loss.backward()`, `clip_grad_norm_`, `torch.optim.AdamW` — runs unchanged on the product modules and yields the
    golden gradients: autograd crosses our Functions and torch's own ops in both directions (non-contiguous, channel-sliced and
    permuted gradient tensors included)."""
    from oracle import losses_ref
    from diffusion_e2e_ft_amd.scheduler import DDIMScheduler
    unet, vae = _models(dev)
    batch, text = gc.train_batch()
    gold = GOLD["depth"]
    sched = DDIMScheduler()
    alpha_prod = sched.alphas_cumprod.to(dev)
    beta_prod = 1 - alpha_prod
    optimizer = torch.optim.AdamW(unet.parameters(), lr=3e-5, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    # ---- train.py:472-500
    with torch.no_grad():
        h = vae.encoder(batch["rgb"].to(dev))
        moments = vae.quant_conv(h)
        rgb_latents, _ = torch.chunk(moments, 2, dim=1)
        rgb_latents = rgb_latents * vae.config.scaling_factor
    val_mask = batch["val_mask"].bool().to(dev)
    timesteps = (torch.ones((rgb_latents.shape[0],), device=dev) * 999).long()
    noisy_latents = torch.zeros_like(rgb_latents)
    encoder_hidden_states = text.to(dev).repeat(len(batch["rgb"]), 1, 1)
    unet_input = torch.cat((rgb_latents, noisy_latents), dim=1)
    model_pred = unet(unet_input, timesteps, encoder_hidden_states, return_dict=False)[0]
    # ---- train.py:509-540
    alpha_prod_t = alpha_prod[timesteps].view(-1, 1, 1, 1)
    beta_prod_t = beta_prod[timesteps].view(-1, 1, 1, 1)
    current_latent_estimate = (alpha_prod_t ** 0.5) * noisy_latents - (beta_prod_t ** 0.5) * model_pred
    current_latent_estimate = current_latent_estimate / vae.config.scaling_factor
    z = vae.post_quant_conv(current_latent_estimate)
    current_estimate = vae.decoder(z)
    current_estimate = current_estimate.mean(dim=1, keepdim=True)
    current_estimate = torch.clamp(current_estimate, -1, 1)
    loss = losses_ref.ssi_loss_ref(current_estimate, batch["metric"].to(dev), val_mask)
    assert abs(loss.item() - gold["loss"].item()) <= 1e-4 * abs(gold["loss"].item())
    # ---- train.py:562-566
    loss.backward()
    _check_grads(dict(unet.named_parameters()), gold)
    before = {k: v.detach().clone() for k, v in unet.named_parameters()}
    torch.nn.utils.clip_grad_norm_(unet.parameters(), 1.0)
    optimizer.step()
    optimizer.zero_grad()
    assert any(not torch.equal(v, before[k]) for k, v in unet.named_parameters())
    with torch.no_grad():   # the kernels see the updated weights (packed copies are keyed on the parameter version)
        out2 = unet(unet_input, timesteps, encoder_hidden_states, return_dict=False)[0]
    assert not torch.equal(out2, model_pred.detach())


@pytest.mark.parametrize("geo", [False, True])
def test_gradient_checkpointing_gives_bit_equal_gradients_and_saves_memory(dev, geo):
    """`unet.enable_gradient_checkpointing()` (training/train.py:342-343): per-block activation recompute as in
    unet_2d_blocks.py:1136-1161 — the loss and every parameter gradient are bit-identical to the run that kept all activations, and the
    peak memory of the step is lower; `vae.enable_gradient_checkpointing()` does the same for the frozen decoder's ResNet blocks"""
    from diffusion_e2e_ft_amd import training
    from diffusion_e2e_ft_amd.unet import UNet2DConditionModel

    def run(ckpt):
        if geo:
            unet = UNet2DConditionModel(**config.TINY_GEOWIZARD_UNET)
            unet.load_state_dict(gc.tiny_geo_sd())
            unet = unet.to(dev).train()
            _, vae = _models(dev)
            batch, emb = gc.geo_train_inputs()
        else:
            unet, vae = _models(dev)
            batch, emb = gc.train_batch(B=4, H=128, W=128)
        if ckpt:
            unet.enable_gradient_checkpointing()
            vae.enable_gradient_checkpointing()
            assert unet.gradient_checkpointing and all(b.gradient_checkpointing for b in unet.down_blocks)
        _pygc.collect()                   # garbage of earlier tests that only the cycle collector frees must not be released INSIDE the measured region
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        loss = (training.geowizard_e2e_ft_loss(unet, vae, batch, emb, "indoor") if geo else training.e2e_ft_loss(unet, vae, batch, emb, "depth"))
        torch.cuda.synchronize()
        held = torch.cuda.memory_allocated() - base      # what the forward pass keeps alive for the backward pass
        loss.backward()
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() - base
        return loss.item(), {k: p.grad.clone() for k, p in unet.named_parameters()}, peak, held

    l0, g0, m0, h0 = run(False)
    l1, g1, m1, h1 = run(True)
    assert l0 == l1
    assert all(torch.equal(g0[k], g1[k]) for k in g0), [k for k in g0 if not torch.equal(g0[k], g1[k])][:5]
    print("kept by the forward pass: %.1f MB / %.1f MB with recompute; peak of the step: %.1f MB / %.1f MB" % (h0 / 2 ** 20, h1 / 2 ** 20, m0 / 2 ** 20, m1 / 2 ** 20))
    # the activations held between forward and backward shrink (at this toy size the PEAK is set by the backward pass's transient
    # weight-gradient operands, which recompute does not touch: bench.py --train --grad-ckpt reports the full-size peaks, 123.7 -> 54.8 GiB)
    assert h1 < 0.8 * h0, (h0, h1)
    assert m1 <= 1.02 * m0, (m0, m1)
    # eval mode: the switch is inert (the reference checks `self.training and self.gradient_checkpointing`)
    unet, vae = _models(dev)
    unet.enable_gradient_checkpointing()
    unet.eval()
    x, ctx = gc.unet_inputs((16, 16))
    with torch.no_grad():
        a = unet(x.to(dev), torch.tensor(999, device=dev), ctx.to(dev)).sample
    unet.disable_gradient_checkpointing()
    with torch.no_grad():
        b = unet(x.to(dev), torch.tensor(999, device=dev), ctx.to(dev)).sample
    assert torch.equal(a, b)


def _toy(dev, seed=0):
    torch.manual_seed(seed)
    net = torch.nn.Sequential(torch.nn.Linear(16, 24), torch.nn.Tanh(), torch.nn.Linear(24, 8), torch.nn.Linear(8, 1)).to(dev)
    return net


def test_flat_adamw_in_the_reference_training_loop_with_lambdalr_and_resume(dev):
    """The optimizer as training/train.py uses it: `torch.optim.AdamW(...)` -> FlatAdamW, `LambdaLR(optimizer, IterExponential)` (:356-357),
    `clip_grad_norm_` + `optimizer.step()` + `lr_scheduler.step()` + `optimizer.zero_grad()` (:561-566), `save_state` / `load_state` (:417-440,
    578-599).  Five steps against torch.optim.AdamW on the same data; then state_dict -> fresh model + optimizer -> load -> the next step must be
    bit-equal to the uninterrupted run."""
    from torch.optim.lr_scheduler import LambdaLR
    from diffusion_e2e_ft_amd.training import FlatAdamW, IterExponential
    net, ref = _toy(dev), _toy(dev)
    lam = IterExponential(total_iter_length=12, final_ratio=0.01, warmup_steps=3)
    opt = FlatAdamW(net.parameters(), lr=2e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=1.0)
    ropt = torch.optim.AdamW(ref.parameters(), lr=2e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    sched, rsched = LambdaLR(opt, lr_lambda=lam), LambdaLR(ropt, lr_lambda=lam)
    g = torch.Generator(device=dev).manual_seed(3)
    xs = [torch.randn(32, 16, device=dev, generator=g) for _ in range(7)]

    def one(model, optimizer, scheduler, x, clip_ref):
        loss = (model(x) ** 2).mean() * 40.0
        loss.backward()
        if clip_ref:
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        optimizer.step()
        scheduler.step()
        optimizer.zero_grad()
        return loss

    for i in range(5):
        one(net, opt, sched, xs[i], False)
        one(ref, ropt, rsched, xs[i], True)
        assert abs(opt.param_groups[0]["lr"] - ropt.param_groups[0]["lr"]) < 1e-15
        for p, rp in zip(net.parameters(), ref.parameters()):
            assert rel_err(p, rp) < 2e-5, i
    assert opt.step_count == 5 and opt.skipped_steps() == 0
    # ---- checkpoint, then the uninterrupted run takes step 6
    ckpt = {"model": {k: v.clone() for k, v in net.state_dict().items()}, "opt": opt.state_dict(), "sched": sched.state_dict()}
    one(net, opt, sched, xs[5], False)
    # ---- resume in fresh objects
    net2 = _toy(dev, seed=99)
    net2.load_state_dict(ckpt["model"])
    opt2 = FlatAdamW(net2.parameters(), lr=123.0, max_grad_norm=1.0)          # hyper-parameters come back from the checkpoint
    sched2 = LambdaLR(opt2, lr_lambda=lam)
    opt2.load_state_dict(ckpt["opt"])
    sched2.load_state_dict(ckpt["sched"])
    assert opt2.step_count == 5 and opt2.param_groups[0]["lr"] == ckpt["opt"]["param_groups"][0]["lr"]
    one(net2, opt2, sched2, xs[5], False)
    for p, p2 in zip(net.parameters(), net2.parameters()):
        assert torch.equal(p, p2)
    assert torch.equal(opt.exp_avg, opt2.exp_avg) and torch.equal(opt.exp_avg_sq, opt2.exp_avg_sq) and opt2.step_count == 6
    # ---- and torch.optim.AdamW resumes from OUR checkpoint to the same step (torch's own state format)
    ref3 = _toy(dev, seed=98)
    ref3.load_state_dict(ckpt["model"])
    ropt3 = torch.optim.AdamW(ref3.parameters(), lr=1.0)
    rsched3 = LambdaLR(ropt3, lr_lambda=lam)          # objects first, then their states (what accelerator.load_state does): a scheduler
    ropt3.load_state_dict({k: v for k, v in ckpt["opt"].items() if k != "flat_adamw"})   # built AFTER the load would reset the group's lr
    rsched3.load_state_dict(ckpt["sched"])
    one(ref3, ropt3, rsched3, xs[5], True)
    for p, rp in zip(net.parameters(), ref3.parameters()):
        assert rel_err(p, rp) < 2e-5


def test_flat_adamw_skips_a_non_finite_gradient_on_the_device_and_counts_it(dev):
    """a NaN / inf gradient norm must not reach the master weights or the moments; the bias-correction step does not advance; the caller can see
    the skip.  Also with max_grad_norm = 0 (no clipping), where the old code path had no guard."""
    from diffusion_e2e_ft_amd.training import FlatAdamW
    for mgn in (1.0, 0.0):
        net = _toy(dev)
        opt = FlatAdamW(net.parameters(), lr=1e-2, max_grad_norm=mgn)
        x = torch.randn(8, 16, device=dev)
        net(x).sum().backward()
        opt.step()
        opt.zero_grad()
        before = [p.detach().clone() for p in net.parameters()]
        m0, v0 = opt.exp_avg.clone(), opt.exp_avg_sq.clone()
        net(x).sum().backward()
        next(net.parameters()).grad[0, 0] = float("nan") if mgn else float("inf")
        opt.step()
        opt.zero_grad()
        assert all(torch.equal(a, b) for a, b in zip(before, net.parameters())) and torch.equal(m0, opt.exp_avg) and torch.equal(v0, opt.exp_avg_sq)
        assert opt.step_count == 1 and opt.skipped_steps() == 1
        net(x).sum().backward()
        opt.step()                                   # a clean step afterwards is step 2, not 3
        assert opt.step_count == 2 and opt.skipped_steps() == 1
        assert not any(torch.equal(a, b) for a, b in zip(before, net.parameters()))
        assert all(torch.isfinite(p).all() for p in net.parameters())


def test_bf16_weights_are_views_of_one_flat_cast_and_steps_are_bit_equal(dev):
    """bf16 compute over fp32 master weights in a FlatAdamW buffer (round 4): convolution weights lie OHWI in the buffer, ONE e2eft_cast per parameter state
    produces the 16-bit twin and the operands of the kernels are views of it.  (i) the views equal what the per-tensor pack / cat / cast path builds, element for
    element; (ii) two optimizer steps with the twin equal two steps of an identical model with the twin switched off (the round-3 per-tensor path) bit for bit."""
    import copy
    from diffusion_e2e_ft_amd import training, autograd as F
    batch, text = gc.train_batch()
    bf = torch.bfloat16

    def make():
        unet, vae = _models(dev)
        unet.set_compute_dtype(bf)
        vae = vae.to(bf)
        return unet, vae, training.FlatAdamW(unet.parameters(), lr=1e-3, max_grad_norm=1.0)

    unet, vae, opt = make()
    twin = opt.shadow.twin(bf)
    lo, hi = twin.data_ptr(), twin.data_ptr() + twin.numel() * 2
    conv = unet.down_blocks[0].resnets[0].conv1
    pk = F.packed_conv_weight(conv, bf)
    assert lo <= pk.data_ptr() < hi and pk.shape == (conv.weight.shape[0], 9 * conv.weight.shape[1]) and pk.is_contiguous()
    assert torch.equal(pk, conv.weight.detach().to(bf).permute(0, 2, 3, 1).reshape(pk.shape))
    att = unet.down_blocks[0].attentions[0].transformer_blocks[0].attn1
    ws = (att.to_q.weight, att.to_k.weight, att.to_v.weight)
    cat = F._cat_weight(att, "wqkv", ws, bf, ws[0].shape[1])
    assert lo <= cat.data_ptr() < hi and torch.equal(cat, torch.cat([w.detach().to(bf) for w in ws]))
    b = F._vec(conv.bias, bf)
    assert lo <= b.data_ptr() < hi and torch.equal(b, conv.bias.detach().to(bf))
    assert F.packed_conv_weight(vae.decoder.conv_in, bf).data_ptr() < lo or F.packed_conv_weight(vae.decoder.conv_in, bf).data_ptr() >= hi     # frozen: its own cache
    vkey = F._key(vae.decoder.conv_in.weight)
    losses = {}
    for tagged in (True, False):
        u, v, o = (unet, vae, opt) if tagged else make()
        F.FLAT_SHADOW_ENABLED = tagged            # False: the per-tensor cast / pack / cat path of round 3
        try:
            ls = []
            for _ in range(2):
                loss = training.e2e_ft_loss(u, v, batch, text, "depth")
                loss.backward()
                o.step()
                o.zero_grad()
                ls.append(loss.item())
            torch.cuda.synchronize()
        finally:
            F.FLAT_SHADOW_ENABLED = True
        losses[tagged] = (ls, o.flat_param.detach().clone())
    assert losses[True][0] == losses[False][0], (losses[True][0], losses[False][0])
    assert torch.equal(losses[True][1], losses[False][1])
    assert F._key(vae.decoder.conv_in.weight) == vkey           # the frozen decoder's packed weights survive optimizer steps (the epoch only concerns flat-resident parameters)


def test_bf16_twin_follows_in_place_parameter_writes(dev):
    """FlatAdamW binds parameters with `p.data = view`, so `p.copy_()` (load_state_dict, an EMA `copy_to`) writes the flat fp32 buffer through the
    PARAMETER's version counter — neither the buffer's `_version` nor the optimizer epoch moves.  The 16-bit twin must still be re-cast: a bf16 forward, then
    `load_state_dict` of different weights, then a forward again must equal a freshly built model with those weights, bit for bit."""
    from diffusion_e2e_ft_amd import training, autograd as F
    batch, text = gc.train_batch()
    bf = torch.bfloat16

    def make(sd):
        unet, vae = _models(dev)
        unet.load_state_dict(sd)
        unet.set_compute_dtype(bf)
        return unet, vae.to(bf), training.FlatAdamW(unet.parameters(), lr=1e-3)

    sd0 = gc.tiny_unet_sd()
    g = torch.Generator().manual_seed(5)
    sd1 = {k: v + 0.05 * v.abs().mean() * torch.randn(v.shape, generator=g) for k, v in sd0.items()}
    unet, vae, opt = make(sd0)
    with torch.no_grad():
        l0 = training.e2e_ft_loss(unet, vae, batch, text, "depth").item()            # builds the twin from sd0
    casts = opt.shadow.state[bf]
    ver = opt.flat_param._version
    unet.load_state_dict(sd1)                                                          # p.copy_() per parameter, under no_grad
    assert opt.flat_param._version == ver and opt.shadow.state[bf] == casts            # the premise: nothing the old key looked at has moved
    conv = unet.down_blocks[0].resnets[0].conv1
    assert torch.equal(F.packed_conv_weight(conv, bf), conv.weight.detach().to(bf).permute(0, 2, 3, 1).reshape(conv.weight.shape[0], -1))
    with torch.no_grad():
        l1 = training.e2e_ft_loss(unet, vae, batch, text, "depth").item()
    fresh, fvae, _ = make(sd1)
    with torch.no_grad():
        want = training.e2e_ft_loss(fresh, fvae, batch, text, "depth").item()
    assert l1 == want and l1 != l0, (l0, l1, want)
    # a single parameter written in place (what an EMA copy_to does parameter by parameter)
    with torch.no_grad():
        conv.bias.copy_(conv.bias * 2 + 1)
    assert torch.equal(F._vec(conv.bias, bf), conv.bias.detach().to(bf))


def test_ema_on_the_flat_buffer(dev):
    """training.EMAModel over FlatAdamW's parameters (GeoWizard's `--use_ema`, train_depth_normal.py:352-353,785-786,843-850): the shadow is one flat buffer,
    `step()` one e2eft_ema_step launch — bit-equal to diffusers' per-tensor update `s -= (1 - decay) * (s - p)` evaluated by torch on clones; `copy_to` puts the
    averaged weights under the model INCLUDING its bf16 twin (a forward equals a fresh model holding the shadow weights), `restore` brings the live ones back."""
    from diffusion_e2e_ft_amd import training
    batch, text = gc.train_batch()
    bf = torch.bfloat16
    unet, vae = _models(dev)
    unet.set_compute_dtype(bf)
    vae = vae.to(bf)
    opt = training.FlatAdamW(unet.parameters(), lr=1e-3, max_grad_norm=1.0)
    ema = training.EMAModel(unet.parameters(), decay=0.8, model_cls=type(unet), model_config=unet.config)
    assert ema._shadow_flat is not None and ema._shadow_flat.numel() == opt.flat_param.numel()
    assert all(s_.shape == p.shape and s_.stride() == p.stride() for s_, p in zip(ema.shadow_params, unet.parameters()))
    ref = [p.detach().clone() for p in unet.parameters()]
    for it in range(1, 5):
        loss = training.e2e_ft_loss(unet, vae, batch, text, "depth")
        loss.backward()
        opt.step()
        opt.zero_grad()
        ema.step(unet.parameters())
        step = max(0, it - 1)
        decay = 0.0 if step <= 0 else min((1 + step) / (10 + step), 0.8)
        assert ema.cur_decay_value == decay
        for r, p in zip(ref, unet.parameters()):
            r.sub_((1 - decay) * (r - p.detach()))
        torch.cuda.synchronize()
        bad = [i for i, (a, b) in enumerate(zip(ema.shadow_params, ref)) if not torch.equal(a, b)]
        assert not bad, (it, bad[:5])
    with torch.no_grad():
        live_loss = training.e2e_ft_loss(unet, vae, batch, text, "depth").item()
    live = opt.flat_param.detach().clone()
    ema.store(unet.parameters())
    ema.copy_to(unet.parameters())
    assert torch.equal(opt.flat_param, ema._shadow_flat)
    with torch.no_grad():
        ema_loss = training.e2e_ft_loss(unet, vae, batch, text, "depth").item()
    fresh, _ = _models(dev)
    with torch.no_grad():
        for p, s_ in zip(fresh.parameters(), ema.shadow_params):
            p.copy_(s_)
    fresh.set_compute_dtype(bf)
    training.FlatAdamW(fresh.parameters(), lr=1e-3)       # the same weight layout (OHWI twin) as the model under test
    with torch.no_grad():
        want = training.e2e_ft_loss(fresh, vae, batch, text, "depth").item()
    assert ema_loss == want and ema_loss != live_loss, (live_loss, ema_loss, want)
    ema.restore(unet.parameters())
    assert torch.equal(opt.flat_param, live)
    with torch.no_grad():
        assert training.e2e_ft_loss(unet, vae, batch, text, "depth").item() == live_loss


def test_ema_frozen_parameters_are_copied_and_restore_survives_a_reordered_list(dev):
    """(i) diffusers' EMAModel.step COPIES a parameter that does not require a gradient (it only averages trainable ones) — the one-launch flat
    form averages every slot, so it must step aside when a parameter of the buffer is frozen; (ii) `restore()` with a parameter list that is not the
    optimizer's own list object order (here: reversed) falls back to per-slot copies instead of failing."""
    from diffusion_e2e_ft_amd import training
    unet, _ = _models(dev)
    opt = training.FlatAdamW(unet.parameters(), lr=1e-3)
    params = list(unet.parameters())
    ema = training.EMAModel(params, decay=0.5)
    frozen = params[3]
    frozen.requires_grad_(False)
    with torch.no_grad():
        opt.flat_param.mul_(1.5)
    ref = [s_.detach().clone() for s_ in ema.shadow_params]
    p1 = [p.detach().clone() for p in params]
    ema.step(params)                       # step 1: decay 0 (the shadow takes the parameters)
    with torch.no_grad():
        opt.flat_param.mul_(0.5)           # the parameters move on, so that averaging and copying differ in step 2
    ema.step(params)                       # step 2: decay = min(2 / 11, 0.5) > 0
    decay = ema.cur_decay_value
    assert 0.0 < decay < 1.0
    torch.cuda.synchronize()
    for i, (s_, r, a, p) in enumerate(zip(ema.shadow_params, ref, p1, params)):
        if p is frozen:
            assert torch.equal(s_, p.detach()), i
        else:
            want = r.clone()
            want.sub_(1.0 * (want - a))                           # step 1
            want.sub_((1 - decay) * (want - p.detach()))          # step 2
            assert torch.equal(s_, want) and not torch.equal(s_, p.detach()), i
    frozen.requires_grad_(True)
    live = opt.flat_param.detach().clone()
    ema.store(params)
    ema.copy_to(params)
    assert not torch.equal(opt.flat_param, live)
    ema.restore(list(params))              # the same parameters through a NEW list object: the flat path
    assert torch.equal(opt.flat_param, live)
    ema.store(params)
    ema.copy_to(params)
    other = [torch.nn.Parameter(torch.empty_like(p)) for p in params]      # parameters that do not live in the buffer: per-slot copies
    ema.restore(other)
    for o, want in zip(other, (opt._slot(live, i) for i in range(len(params)))):
        assert torch.equal(o.detach(), want)


@pytest.mark.parametrize("cdt,ckpt", [(torch.bfloat16, False), (torch.float32, False), (torch.bfloat16, True)], ids=["bf16", "fp32", "bf16_recompute"])
def test_gradients_are_born_in_the_flat_buffer_and_steps_are_bit_equal(dev, cdt, ckpt):
    """FlatAdamW(direct_grads=True) (round 4): after zero_grad() every .grad is None, the backward kernels' reductions write each parameter's first gradient of
    the step into its slot of the flat fp32 buffer and autograd keeps that view as .grad — no AccumulateGrad add, no memset.  (i) after a backward every gradient
    IS its slot; (ii) optimizer steps — one plain, one with two accumulation micro-steps — equal the same steps with the sink switched off (gradients handed to
    autograd as fresh tensors and adopted) and with direct_grads=False (memset + in-place accumulation, the round-3 behaviour) bit for bit."""
    from diffusion_e2e_ft_amd import training, autograd as F
    batch, text = gc.train_batch()

    def make(direct):
        unet, vae = _models(dev)
        if cdt != torch.float32:
            unet.set_compute_dtype(cdt)
            vae = vae.to(cdt)
        if ckpt:       # activation recompute: the backward kernels run inside torch.utils.checkpoint's re-entered segments
            unet.enable_gradient_checkpointing()
            vae.enable_gradient_checkpointing()
        return unet, vae, training.FlatAdamW(unet.parameters(), lr=1e-3, max_grad_norm=1.0, direct_grads=direct)

    def run(u, v, o, sink):
        F.GRAD_SINK_ENABLED = sink
        try:
            o.zero_grad()
            ls = []
            for n in (1, 2, 1):
                for i in range(n):
                    loss = training.e2e_ft_loss(u, v, batch, text, "depth")
                    (loss / n).backward()
                    ls.append(loss.item())
                if n == 1 and sink and o.direct_grads:
                    born = [q.grad is not None and q.grad.data_ptr() == o.flat_grad.data_ptr() + 4 * off and q.grad.stride() == q.stride()
                            for q, off in zip(o.params, o.offsets)]
                    assert all(born), "%d of %d gradients were not written into their slot: %s" % (
                        len(born) - sum(born), len(born), [k for k, q in u.named_parameters() if any(q is r and not b for r, b in zip(o.params, born))][:8])
                o.step()
                o.zero_grad()
            torch.cuda.synchronize()
        finally:
            F.GRAD_SINK_ENABLED = True
        return ls, o.flat_param.detach().clone()

    a = run(*make(True), True)
    b = run(*make(True), False)
    c = run(*make(False), True)
    assert a[0] == b[0] == c[0], (a[0], b[0], c[0])
    assert torch.equal(a[1], b[1]) and torch.equal(a[1], c[1])
