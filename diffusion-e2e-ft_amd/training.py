"""Host side of the E2E-FT training step (training/train.py:470-568), built on the libe2eft forward/backward kernels.

  e2e_ft_loss        — the step's forward half: frozen VAE encode -> zeros latent -> UNet @ t=999 -> x0 -> frozen VAE decode ->
                       depth / normal head -> ScaleAndShiftInvariantLoss / AngularLoss   (train.py:472-556)
  FlatAdamW          — clip_grad_norm_ + torch.optim.AdamW semantics (train.py:346-353,561-566) on ONE flat fp32 parameter
                       buffer: parameters and their .grad are views of two big allocations, so the data-parallel gradient
                       exchange is a handful of large RCCL all-reduces over contiguous slices (no bucket copies), launched
                       from autograd hooks while the rest of the backward is still running, and the update is one launch.
  IterExponential    — training/util/lr_scheduler.py:10-36
  replace_unet_conv_in — training/util/unet_prep.py:6-20

The reference wraps the UNet in DDP via accelerate; here one process per GPU (torchrun) calls FlatAdamW.step(), which waits for
the outstanding all-reduces.  xGMI is point-to-point, so few large collectives (default 4 byte-sized slices of the 3.46 GB
gradient, the last-finishing one the smallest) keep all seven links busy; the first slice to finish its backward (the up blocks)
starts its exchange while the down blocks are still computing.
"""
import math
import weakref

import torch
import torch.distributed as dist

from . import autograd as F
from . import ops


class IterExponential:
    """lr multiplier: linear warm-up, then exp decay to final_ratio at total_iter_length (lr_scheduler.py:10-36)"""

    def __init__(self, total_iter_length, final_ratio, warmup_steps=0):
        self.total_length = total_iter_length
        self.effective_length = total_iter_length - warmup_steps
        self.final_ratio = final_ratio
        self.warmup_steps = warmup_steps

    def __call__(self, n_iter):
        if n_iter < self.warmup_steps:
            return 1.0 * n_iter / self.warmup_steps
        if n_iter >= self.total_length:
            return self.final_ratio
        return math.exp((n_iter - self.warmup_steps) / self.effective_length * math.log(self.final_ratio))


def replace_unet_conv_in(unet, repeat=2):
    """4 -> 8 input channels: repeat the weight on Cin, halve weight and bias (unet_prep.py:6-20)"""
    w = unet.conv_in.weight.detach().clone().repeat(1, repeat, 1, 1) / repeat
    b = unet.conv_in.bias.detach().clone() / repeat
    old = unet.conv_in
    new = type(old)(old.in_channels * repeat, old.out_channels, kernel_size=old.kernel_size, stride=old.stride, padding=old.padding)
    new = new.to(device=w.device, dtype=w.dtype)
    with torch.no_grad():
        new.weight.copy_(w)
        new.bias.copy_(b)
    unet.conv_in = new
    unet.config["in_channels"] = old.in_channels * repeat


# ------------------------------------------------------------------------------------------------------------
def encode_image(vae, image):
    """train.py:233-237"""
    h = vae.encoder(image)
    moments = vae.quant_conv(h)
    latent, _ = torch.chunk(moments, 2, dim=1)
    return latent


def e2e_ft_loss(unet, vae, batch, empty_encoding, modality="depth", noise_scheduler=None, return_estimate=False):
    """Forward half of one micro-step (train.py:472-556) with the zeros latent at t = 999 (the E2E-FT recipe).
    batch: rgb [b,3,H,W] in [-1,1], val_mask [b,1,H,W] bool, metric [b,1,H,W] / normals [b,3,H,W].  Returns the scalar loss
    (device tensor with a grad_fn through the decoder and the UNet).  `unet` may be wrapped (DistributedDataParallel by
    `accelerator.prepare`, train.py:369): attributes are read from `.module`, the forward goes through the wrapper."""
    core = getattr(unet, "module", unet)
    dev = core.device
    with ops.on_device_of(next(core.parameters())):
        return _e2e_ft_loss(unet, core, vae, batch, empty_encoding, modality, noise_scheduler, return_estimate)


def _e2e_ft_loss(unet, core, vae, batch, empty_encoding, modality, noise_scheduler, return_estimate):
    dev = core.device
    dt = getattr(core, "compute_dtype", core.dtype)
    with torch.no_grad():
        rgb_latents = encode_image(vae, batch["rgb"].to(device=dev, dtype=dt)) * vae.config.scaling_factor
    val_mask = batch["val_mask"].bool().to(dev)
    b = rgb_latents.shape[0]
    timesteps = torch.full((b,), 999, device=dev, dtype=torch.long)
    noisy = torch.zeros_like(rgb_latents)
    ctx = empty_encoding.to(device=dev, dtype=dt).repeat(b, 1, 1)
    unet_input = torch.cat((rgb_latents, noisy), dim=1).contiguous(memory_format=torch.channels_last)
    model_pred = unet(unet_input, timesteps, ctx, return_dict=False)[0]
    if noise_scheduler is None:
        from .scheduler import DDIMScheduler
        noise_scheduler = DDIMScheduler()
    # x_t = 0: x0 = c * model_pred with c by the scheduler's prediction_type (v: -sqrt(1 - alpha_prod), train.py:509-518), then / scaling_factor (:528)
    x0 = model_pred * (noise_scheduler.zero_latent_x0_scale(999) / vae.config.scaling_factor)
    est = vae.decoder(vae.post_quant_conv(x0))                     # [b,3,H,W] logical NCHW, NHWC memory
    est_nhwc = est.permute(0, 2, 3, 1)
    if modality == "depth":
        cur = F.depth_head(est_nhwc, to_unit=False)                # mean over channels, clamp(-1, 1)  (:531-533)
        loss = F.ssi_loss(cur, batch["metric"].to(device=dev), val_mask)
    elif modality == "normals":
        cur = F.normal_head(est_nhwc, clamp=True)                  # x / (|x| + 1e-5), clamp  (:535-538)
        loss = F.angular_loss(cur, batch["normals"].to(device=dev), val_mask)
    else:
        raise ValueError("Unknown modality %s" % modality)
    return (loss, cur) if return_estimate else loss


def geowizard_class_embedding(batch, domain, dtype, device):
    """hybrid switcher + domain class, rows [depth (b); normal (b)]  (train_depth_normal.py:684-700)"""
    geo = torch.tensor([[0.0, 1.0], [1.0, 0.0]], dtype=torch.float32, device=device).repeat_interleave(batch, 0)
    dom = {"indoor": [1.0, 0.0, 0.0], "outdoor": [0.0, 1.0, 0.0], "object": [0.0, 0.0, 1.0]}[domain]
    dom = torch.tensor([dom], dtype=torch.float32, device=device).repeat(2 * batch, 1)
    return torch.cat([torch.sin(geo), torch.cos(geo), torch.sin(dom), torch.cos(dom)], dim=-1).to(dtype)


def geowizard_e2e_ft_loss(unet, vae, batch, imgs_embed, domain="indoor", depth_scale=0.5, normal_scale=1.0, return_parts=False):
    """Forward half of the GeoWizard E2E-FT micro-step (GeoWizard/geowizard/training/train_depth_normal.py:597-768, --e2e_ft, zeros
    noise): the UNet runs the doubled batch [depth rows; normal rows] with cross-domain joint self-attention and the class
    embedding, the frozen decoder decodes both halves, loss = 0.5 * SSI(depth) + 1.0 * angular(normals vs -GT).
    imgs_embed: CLIP image embeddings [b,1,768] (an input here)."""
    core = getattr(unet, "module", unet)
    with ops.on_device_of(next(core.parameters())):
        return _geowizard_e2e_ft_loss(unet, core, vae, batch, imgs_embed, domain, depth_scale, normal_scale, return_parts)


def _geowizard_e2e_ft_loss(unet, core, vae, batch, imgs_embed, domain, depth_scale, normal_scale, return_parts):
    dev = core.device
    dt = getattr(core, "compute_dtype", core.dtype)
    with torch.no_grad():
        rgb_latents = encode_image(vae, batch["rgb"].to(device=dev, dtype=dt)) * vae.config.scaling_factor
    val_mask = batch["val_mask"].bool().to(dev)
    b = rgb_latents.shape[0]
    timesteps = torch.full((2 * b,), 999, device=dev, dtype=torch.long)
    noisy = torch.zeros_like(rgb_latents).repeat(2, 1, 1, 1)
    ctx = imgs_embed.to(device=dev, dtype=dt).repeat(2, 1, 1)
    cls = geowizard_class_embedding(b, domain, dt, dev)
    unet_input = torch.cat((rgb_latents.repeat(2, 1, 1, 1), noisy), dim=1).contiguous(memory_format=torch.channels_last)
    noise_pred = unet(unet_input, timesteps, ctx, class_labels=cls, return_dict=False)[0]
    from .scheduler import DDIMScheduler
    x0 = noise_pred * (DDIMScheduler().zero_latent_x0_scale(999) / vae.config.scaling_factor)
    est = vae.decoder(vae.post_quant_conv(x0)).permute(0, 2, 3, 1)          # NHWC view [2b,H,W,3]
    depth = F.depth_head(est[:b], to_unit=False)
    normal = F.normal_head(est[b:], clamp=True)
    ssi = F.ssi_loss(depth, batch["metric"].to(device=dev), val_mask)
    ang = F.angular_loss(normal, batch["normals"].to(device=dev) * -1, val_mask)     # GeoWizard trains on inverted normals (:611,751)
    loss = ssi * depth_scale + ang * normal_scale
    return (loss, ssi, ang) if return_parts else loss


# ------------------------------------------------------------------------------------------------------------
class FlatAdamW(torch.optim.Optimizer):
    """`torch.optim.AdamW` + `clip_grad_norm_` (training/train.py:346-353,561-566) over ONE flat fp32 buffer, with the data-parallel
    gradient all-reduce overlapped with the backward.

    A real `torch.optim.Optimizer`: one `param_groups` entry whose `lr` is live (so `LambdaLR(optimizer, IterExponential(...))` of
    train.py:356-357 drives it), `state[p] = {"step", "exp_avg", "exp_avg_sq"}` per parameter (views of the flat moment buffers and of
    the device-side step counter), `state_dict()` / `load_state_dict()` in torch's own format (`accelerator.save_state / load_state`,
    train.py:417-440,578-599, and checkpoints written by `torch.optim.AdamW` load here and vice versa), `zero_grad(set_to_none=...)`,
    `step(closure=None)`.

    What is flat: the storage of every trainable Parameter is moved into `self.flat_param` (diffusers-layout views: `state_dict()` of the
    model keeps working), `.grad` is pre-bound to views of `self.flat_grad`, so autograd accumulates straight into the exchange buffer,
    the exchange is a handful of large RCCL all-reduces over contiguous slices (no bucket copies) launched from autograd hooks while the
    rest of the backward is still running, and clip + update are one `sumsq` and one `adamw` launch.

    external_grad_sync=True: somebody else averages the gradients across ranks (a DistributedDataParallel wrap by `accelerator.prepare`,
    train.py:369) — no hooks, no division by the world size here.
    Non-finite gradient norm: the step is skipped on the device, the bias-correction step does not advance, `skipped_steps()` counts it
    (e2eft_adamw_step_guarded, include/e2eft.h).

    direct_grads=True (default): `zero_grad()` leaves every `.grad` None (torch's own default) instead of clearing the flat buffer, and the backward kernels of
    the libe2eft autograd Functions write a parameter's FIRST gradient of an optimizer step straight into its slot of `flat_grad` (autograd.grad_sink): the
    tensor autograd then stores as `.grad` is a view of the exchange buffer, no AccumulateGrad add per parameter, no 3.5 GB memset per step.  Later gradients of the
    same step (gradient accumulation) are accumulated by autograd in place as before; gradients that arrive any other way are adopted (copied) into their slot by
    the hook / `step()`; slots that received nothing are zeroed there.  Scope of the sink: `loss.backward()` accumulating into `.grad`, gradients cleared by
    `zero_grad()` (this optimizer's or the model's, set_to_none=True).  The tensor such a backward hands to autograd is a VIEW of the exchange buffer that later
    steps overwrite in place: do not keep it across steps (`torch.autograd.grad(...)` results, hooks that retain the gradient, a `retain_graph` second backward —
    use direct_grads=False for those)."""

    def __init__(self, params, lr=3e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=1.0, n_slices=4,
                 process_group=None, external_grad_sync=False, direct_grads=True):
        params = list(params)
        if params and isinstance(params[0], dict):
            if len(params) != 1:
                raise ValueError("FlatAdamW updates one flat buffer with one set of hyper-parameters: pass a single parameter group")
            group0 = dict(params[0])
            plist = [p for p in group0.pop("params") if p.requires_grad]
            lr, betas, eps, weight_decay = group0.get("lr", lr), group0.get("betas", betas), group0.get("eps", eps), group0.get("weight_decay", weight_decay)
        else:
            plist = [p for p in params if p.requires_grad]
        if not plist:
            raise ValueError("no trainable parameters")
        dev = plist[0].device
        for p in plist:
            if p.dtype != torch.float32:
                raise TypeError("FlatAdamW keeps fp32 master parameters; got %s" % p.dtype)
            if p.device != dev:
                raise ValueError("FlatAdamW: all parameters must live on one device")
        super().__init__(plist, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.params = plist
        self.max_grad_norm = max_grad_norm
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 7) // 8 * 8                     # keep every view 16-byte aligned in the 16-bit twin too (autograd.FlatShadow): 32 bytes here
        self.numel = n
        # Convolution weights are stored in the order the kernels read them — OHWI, i.e. the Parameter becomes a channels_last view of its slot (same values, same
        # logical OIHW shape: state_dict / save_pretrained / copy_ see no difference) — so that with 16-bit compute the packed weight of every convolution is a
        # slice of ONE flat cast of this buffer (autograd.FlatShadow) and the weight gradient, which the kernels produce in OHWI order, is accumulated without a
        # permuting copy.
        self._ohwi = [p.dim() == 4 and p.shape[2] * p.shape[3] > 1 for p in plist]
        with ops.on_device_of(plist[0]):
            self.flat_param = torch.zeros(n, dtype=torch.float32, device=dev)
            self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
            self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
            self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
            self._sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
            self._steps = torch.zeros(2, dtype=torch.int64, device=dev)      # {applied, skipped} — advanced by the update kernel itself
            self._coef = torch.zeros(4, dtype=torch.float32, device=dev)
            with torch.no_grad():
                self.shadow = F.FlatShadow(self.flat_param)
                for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                    view = self._slot(self.flat_param, i)
                    view.copy_(p.data)
                    p.data = view
                    p.grad = self._slot(self.flat_grad, i)
                    p._e2eft_flat = (self.shadow, o)
                    p._e2eft_gslot = (weakref.ref(self), i)      # (weak: the tag must not tie the optimizer's buffers into a reference cycle with its parameters)
                self.shadow.register(self.params, self.offsets)
        self.direct_grads = bool(direct_grads)
        self._claimed = [False] * len(self.params)
        self._bind_state()
        F.bump_param_epoch()
        self.group = process_group
        self.world = 1 if external_grad_sync else (dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1)
        # Slices of the flat buffer, cut by BYTES.  Parameters are stored in module order (conv_in, down, mid, up, conv_out) and the
        # backward produces gradients roughly in reverse, so the slice holding the FIRST parameters completes last and its exchange is
        # the one that cannot hide under the backward: slice sizes grow geometrically (1 : 2 : 4 : ...), the exposed one is the smallest
        # (n_slices = 4 on the 3.46 GB SD-v2 gradient: 0.23 / 0.46 / 0.92 / 1.85 GB).  Large messages keep all seven xGMI links busy.
        n_slices = max(1, min(n_slices, len(self.params)))
        total_w = float(2 ** n_slices - 1)
        targets, acc_w = [], 0.0
        for s_ in range(n_slices - 1):
            acc_w += 2 ** s_
            targets.append(n * acc_w / total_w)
        cuts, ti = [0], 0
        for i in range(1, len(self.params)):
            if ti < len(targets) and self.offsets[i] >= targets[ti] and i > cuts[-1]:
                cuts.append(i)
                ti += 1
                while ti < len(targets) and self.offsets[i] >= targets[ti]:
                    ti += 1
        cuts.append(len(self.params))
        self.slices = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            start = self.offsets[lo]
            end = self.offsets[hi] if hi < len(self.params) else n
            self.slices.append(dict(lo=lo, hi=hi, start=start, end=end, ready=0, work=None, done=False))
        self._index_of = {id(p): i for i, p in enumerate(self.params)}
        self._slice_of = {}
        for si, sl in enumerate(self.slices):
            for i in range(sl["lo"], sl["hi"]):
                self._slice_of[id(self.params[i])] = si
        self.sync_grads = True          # set False on non-final gradient-accumulation micro-steps
        self.profile_exchange = False   # bench.py --gpus N: time what step() waits for, per slice (exchange_exposed_ms)
        self._exposed = []
        self._hooks = []
        if self.world > 1:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _slot(self, buf, i):
        """parameter i's view of a flat buffer (parameters, gradients, both moments): its shape, OHWI-ordered for convolution weights"""
        p, o = self.params[i], self.offsets[i]
        flat = buf[o:o + p.numel()]
        if self._ohwi[i]:
            Co, Ci, kh, kw = p.shape
            return flat.view(Co, kh, kw, Ci).permute(0, 3, 1, 2)
        return flat.view(p.shape)

    def _claim(self, idx, params):
        """autograd.grad_sink: hand the gradient slots `idx` (of `params`) to a backward kernel as its output — once per optimizer step and parameter.
        Several slots must lie back to back (q | k | v, LayerNorm weight | bias).  -> (flat fp32 view over the slots, [one view per parameter]) or None"""
        if not self.direct_grads:
            return None
        for k, i in enumerate(idx):
            if self._claimed[i] or self.params[i] is not params[k]:
                return None
            if k > 0 and self.offsets[i] != self.offsets[idx[k - 1]] + self.params[idx[k - 1]].numel():
                return None
        for i in idx:
            self._claimed[i] = True
        lo, hi = self.offsets[idx[0]], self.offsets[idx[-1]] + self.params[idx[-1]].numel()
        return self.flat_grad[lo:hi], [self._slot(self.flat_grad, i) for i in idx]

    # ---- torch.optim.Optimizer surface -------------------------------------------------------------------------------------------
    def _bind_state(self):
        """state[p]: views of the flat moment buffers; "step" is a view of the device-side applied-step counter shared by all parameters"""
        step = self._steps[0]
        for i, p in enumerate(self.params):
            self.state[p] = {"step": step, "exp_avg": self._slot(self.exp_avg, i), "exp_avg_sq": self._slot(self.exp_avg_sq, i)}

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, v):
        self.param_groups[0]["lr"] = v

    @property
    def step_count(self):
        """applied steps (host value; synchronises)"""
        return int(self._steps[0].item())

    def skipped_steps(self):
        """optimizer steps dropped because the gradient norm was not finite (host value; synchronises)"""
        return int(self._steps[1].item())

    def state_dict(self):
        sd = super().state_dict()
        # torch's format wants per-parameter tensors that survive on their own.  super().state_dict() hands out the SAME per-parameter
        # dicts self.state holds, so build new ones: rebinding entries in place would cut state[p] loose from the flat moment buffers
        # and every later checkpoint would be stale.
        sd["state"] = {k: {"step": st["step"].detach().to(torch.float32).clone(), "exp_avg": st["exp_avg"].detach().clone(),
                           "exp_avg_sq": st["exp_avg_sq"].detach().clone()} for k, st in sd["state"].items()}
        sd["flat_adamw"] = {"skipped_steps": self.skipped_steps(), "max_grad_norm": self.max_grad_norm}
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        extra = state_dict.get("flat_adamw", {})
        super().load_state_dict({k: v for k, v in state_dict.items() if k != "flat_adamw"})   # group hyper-parameters (lr, betas, ...) + per-parameter copies
        steps = set()
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            st = self.state.get(p)
            n = p.numel()
            if not st:                            # parameter without saved state (fresh): zero moments
                self.exp_avg[o:o + n].zero_()
                self.exp_avg_sq[o:o + n].zero_()
                continue
            self._slot(self.exp_avg, i).copy_(st["exp_avg"])
            self._slot(self.exp_avg_sq, i).copy_(st["exp_avg_sq"])
            steps.add(int(torch.as_tensor(st["step"]).item()))
        if len(steps) > 1:
            raise ValueError("FlatAdamW.load_state_dict: parameters carry different step counts %s; one flat buffer has one bias correction" % sorted(steps))
        self._steps[0] = steps.pop() if steps else 0
        self._steps[1] = int(extra.get("skipped_steps", 0))
        self._bind_state()

    # autograd hook: a parameter's gradient is final for this backward
    def _on_grad(self, p):
        if not self.sync_grads:
            return
        self._adopt_one(p, self.offsets[self._index_of[id(p)]])      # a re-bound .grad goes back into its slot BEFORE the slice is exchanged
        sl = self.slices[self._slice_of[id(p)]]
        sl["ready"] += 1
        if sl["ready"] == sl["hi"] - sl["lo"]:
            sl["work"] = dist.all_reduce(self.flat_grad[sl["start"]:sl["end"]], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _finish_exchange(self):
        """wait for the slices the hooks launched, exchange the ones no hook completed (parameters without a gradient this step).  Idempotent
        within an optimizer step: a slice is summed over the ranks exactly once until `step()` / `zero_grad()` re-arm it."""
        if self.world == 1:
            return
        for sl in self.slices:
            if sl["work"] is None and not sl["done"]:
                sl["work"] = dist.all_reduce(self.flat_grad[sl["start"]:sl["end"]], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        prof = self.profile_exchange and self.flat_grad.is_cuda
        for si, sl in enumerate(self.slices):
            if sl["work"] is not None:
                if prof:     # how long the compute stream stands still for this slice: events on it either side of the wait
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    sl["work"].wait()
                    e1.record()
                    self._exposed.append((si, e0, e1))
                else:
                    sl["work"].wait()
            sl["work"], sl["ready"], sl["done"] = None, 0, True

    def exchange_exposed_ms(self):
        """profile_exchange = True: per slice, the mean time (ms) the compute stream waited for that slice's all-reduce in `step()` since the last
        call — the part of the exchange the backward did not hide.  Synchronises."""
        if not self._exposed:
            return []
        torch.cuda.synchronize(self.flat_grad.device)
        acc = [[0.0, 0] for _ in self.slices]
        for si, e0, e1 in self._exposed:
            acc[si][0] += e0.elapsed_time(e1)
            acc[si][1] += 1
        self._exposed = []
        return [{"slice": i, "mbytes": (sl["end"] - sl["start"]) * 4 / 1e6, "exposed_ms": a[0] / max(a[1], 1)} for i, (sl, a) in enumerate(zip(self.slices, acc))]

    def _rearm_exchange(self):
        for sl in self.slices:
            sl["done"] = False

    def _adopt_one(self, p, o):
        g = p.grad
        slot = self._slot(self.flat_grad, self._index_of[id(p)])
        if g is not None and g.data_ptr() == slot.data_ptr() and g.stride() == slot.stride():
            return
        if g is None:
            slot.zero_()
        else:
            slot.copy_(g)
        p.grad = slot

    def _adopt_grads(self):
        """A caller may have re-bound .grad (zero_grad(set_to_none=True) of a generic training loop, then autograd created fresh tensors;
        a DDP wrap handing out its bucket views): bring every gradient back into its slot of the flat buffer.  No-op in the normal case.
        Runs BEFORE the exchange (the hooks adopt the parameter they fire for; `step()` adopts the rest, i.e. parameters that got no
        gradient, whose slice no hook completed and which is therefore exchanged afterwards by `_finish_exchange`)."""
        for p, o in zip(self.params, self.offsets):
            self._adopt_one(p, o)

    @torch.no_grad()
    def step(self, closure=None, lr_scale=1.0, grad_scale=1.0):
        """all-reduce(mean) -> clip_grad_norm_(max_grad_norm) -> AdamW, no host synchronisation.  lr = param_groups[0]["lr"] * lr_scale."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        with ops.on_device_of(self.flat_param):
            self._adopt_grads()
            self._finish_exchange()
            g = self.param_groups[0]
            sumsq = ops.sumsq(self.flat_grad, out=self._sumsq)     # always: the non-finite guard needs it even without clipping
            ops.adamw_step_guarded_(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, float(g["lr"]) * lr_scale, g["betas"][0], g["betas"][1],
                                    g["eps"], g["weight_decay"], self._steps, self._coef, sumsq, grad_scale=grad_scale / self.world,
                                    max_norm=float(self.max_grad_norm or 0.0))
        self._rearm_exchange()
        self._claimed = [False] * len(self.params)       # a loop that clears gradients with model.zero_grad() instead of ours still gets the sink next step
        F.bump_param_epoch()
        return loss

    def grad_norm(self):
        """global L2 norm of the (averaged) gradient — host value, synchronises.  With world > 1 and `sync_grads` this completes the exchange (each slice
        is summed once per optimizer step, so a following `step()` does not exchange again).  On a gradient-accumulation micro-step (`sync_grads` False) nothing
        is exchanged — summing a partial accumulation across ranks early and then adding local micro-gradients on top would corrupt the step —
        and the value is the norm of THIS rank's local gradient so far."""
        with ops.on_device_of(self.flat_param):
            self._adopt_grads()
            if not self.sync_grads:
                return math.sqrt(float(ops.sumsq(self.flat_grad, out=self._sumsq).item()))
            self._finish_exchange()
            return math.sqrt(float(ops.sumsq(self.flat_grad, out=self._sumsq).item())) / self.world

    @torch.no_grad()
    def zero_grad(self, set_to_none=None):
        """direct_grads (the default; or set_to_none=True): every `.grad` becomes None and the flat buffer is NOT cleared — the next backward's kernels write
        the gradients into their slots (see the class docstring), `step()` zeroes the slots that got none.  Otherwise (direct_grads=False, or set_to_none=False
        asked explicitly): ONE memset of the flat buffer, `.grad` stays bound to its slot and autograd accumulates into it."""
        for sl in self.slices:                           # a backward without a step() (skipped iteration): drain its exchange first,
            if sl["work"] is not None:                   # the buffer must not be rewritten under an in-flight all-reduce
                sl["work"].wait()
            sl["work"], sl["ready"] = None, 0
        self._rearm_exchange()
        self._claimed = [False] * len(self.params)
        if self.direct_grads if set_to_none is None else set_to_none:
            for p in self.params:
                p.grad = None
            return
        self.flat_grad.zero_()
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):      # autograd may have re-bound .grad (e.g. set_to_none by a caller)
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self._slot(self.flat_grad, i)


class EMAModel:
    """`diffusers.training_utils.EMAModel` as GeoWizard's training script uses it (GeoWizard/geowizard/training/train_depth_normal.py:352-353 construct, :785-786
    `step` after every optimizer step, :843-850 `store` / `copy_to` / `restore` around validation, :380-381,388-391 `save_pretrained` / `from_pretrained` /
    `load_state_dict` in the accelerate hooks).  diffusers (0.30.2, third-party) is not in the tree: the constructor arguments, `get_decay`, `step`'s update
    `shadow -= (1 - decay) * (shadow - param)` and the `state_dict` keys are restated from its published definition (tests pin the arithmetic to that
    formula evaluated by torch).

    MI355X-first: when the parameters live in a FlatAdamW buffer (construct the optimizer first) the shadow is ONE flat fp32 buffer with the same layout and
    `step()` is ONE `e2eft_ema_step` launch over it (12 B per parameter, HBM-bound: 10.4 GB -> ~2 ms for the SD-v2 UNet, against 686 x 3 elementwise torch launches);
    `copy_to` / `store` / `restore` are flat device copies (the reference parks the stored copy on the host: 288 GB of HBM make that unnecessary) and bump the
    parameter epoch so that 16-bit weight twins and packed weights are rebuilt.  `shadow_params` stays a list of per-parameter tensors (views of the flat
    shadow).  Parameters outside a flat buffer take the per-tensor form of the same update (torch ops: host-side plumbing, as diffusers does it)."""

    def __init__(self, parameters, decay=0.9999, min_decay=0.0, update_after_step=0, use_ema_warmup=False, inv_gamma=1.0, power=2 / 3, foreach=False,
                 model_cls=None, model_config=None, **kwargs):
        parameters = list(parameters)
        self.decay, self.min_decay, self.update_after_step = decay, min_decay, update_after_step
        self.use_ema_warmup, self.inv_gamma, self.power = use_ema_warmup, inv_gamma, power
        self.optimization_step = 0
        self.cur_decay_value = None
        self.temp_stored_params = None
        self.model_cls, self.model_config = model_cls, model_config
        self._flat = self._flat_source(parameters)
        if self._flat is not None:
            owner, flat = self._flat
            with ops.on_device_of(flat):
                self._shadow_flat = flat.detach().clone()
            self.shadow_params = [owner._slot(self._shadow_flat, i) for i in range(len(owner.params))]
        else:
            self._shadow_flat = None
            self.shadow_params = [p.clone().detach() for p in parameters]

    @staticmethod
    def _flat_source(parameters):
        """(FlatAdamW, its flat parameter buffer) when `parameters` are exactly that optimizer's parameters, in its order; else None"""
        tags = [getattr(p, "_e2eft_gslot", None) for p in parameters]
        if not parameters or any(t is None for t in tags):
            return None
        owner = tags[0][0]()
        if owner is None or len(owner.params) != len(parameters) or any(a is not b for a, b in zip(owner.params, parameters)):
            return None
        if any(p.data_ptr() != owner.flat_param.data_ptr() + 4 * o for p, o in zip(parameters, owner.offsets)):
            return None
        return owner, owner.flat_param

    def get_decay(self, optimization_step):
        step = max(0, optimization_step - self.update_after_step - 1)
        if step <= 0:
            return 0.0
        cur = 1 - (1 + step / self.inv_gamma) ** -self.power if self.use_ema_warmup else (1 + step) / (10 + step)
        return max(min(cur, self.decay), self.min_decay)

    @torch.no_grad()
    def step(self, parameters):
        parameters = list(parameters)
        self.optimization_step += 1
        decay = self.get_decay(self.optimization_step)
        self.cur_decay_value = decay
        omd = 1 - decay
        src = self._flat_source(parameters) if self._shadow_flat is not None else None
        # the one-launch form updates EVERY slot; diffusers copies a parameter that does not require a gradient instead of averaging it (below):
        # the flat kernel therefore runs only when every parameter of the buffer is trainable
        if src is not None and src[1].numel() == self._shadow_flat.numel() and all(p.requires_grad for p in parameters):
            with ops.on_device_of(src[1]):
                ops.ema_step_(self._shadow_flat, src[1], omd)
            return
        for s_param, param in zip(self.shadow_params, parameters):
            if param.requires_grad:
                s_param.sub_(omd * (s_param - param))
            else:
                s_param.copy_(param)

    @torch.no_grad()
    def copy_to(self, parameters):
        parameters = list(parameters)
        src = self._flat_source(parameters) if self._shadow_flat is not None else None
        if src is not None and src[1].numel() == self._shadow_flat.numel():
            src[1].copy_(self._shadow_flat)
        else:
            for s_param, param in zip(self.shadow_params, parameters):
                param.data.copy_(s_param.to(param.device).data)
        F.bump_param_epoch()          # (`.data.copy_` moves no version counter: derived 16-bit / packed weights are keyed on the epoch)

    @torch.no_grad()
    def store(self, parameters):
        parameters = list(parameters)
        src = self._flat_source(parameters) if self._shadow_flat is not None else None
        self.temp_stored_params = src[1].detach().clone() if src is not None else [p.detach().clone() for p in parameters]

    @torch.no_grad()
    def restore(self, parameters):
        if self.temp_stored_params is None:
            raise RuntimeError("This ExponentialMovingAverage has no `store()`ed weights to `restore()`")
        parameters = list(parameters)
        if isinstance(self.temp_stored_params, torch.Tensor):
            src = self._flat_source(parameters)
            if src is not None and src[1].numel() == self.temp_stored_params.numel():
                src[1].copy_(self.temp_stored_params)
            else:   # `parameters` are not (any more) the optimizer's list in its order: per-slot copies out of the stored flat buffer, in the stored order
                owner = self._flat[0]
                if len(parameters) != len(owner.params):
                    raise ValueError("restore(): %d parameters were stored, %d given" % (len(owner.params), len(parameters)))
                for i, param in enumerate(parameters):
                    param.data.copy_(owner._slot(self.temp_stored_params, i))
        else:
            for c_param, param in zip(self.temp_stored_params, parameters):
                param.data.copy_(c_param.data)
        self.temp_stored_params = None
        F.bump_param_epoch()

    def to(self, device=None, dtype=None):
        if self._shadow_flat is not None:
            if device is not None and torch.device(device) != self._shadow_flat.device or dtype not in (None, torch.float32):
                raise ValueError("EMAModel over a FlatAdamW buffer lives where the buffer lives (fp32 on %s)" % self._shadow_flat.device)
            return
        self.shadow_params = [p.to(device=device, dtype=dtype) if p.is_floating_point() else p.to(device=device) for p in self.shadow_params]

    def state_dict(self):
        return {"decay": self.decay, "min_decay": self.min_decay, "optimization_step": self.optimization_step, "update_after_step": self.update_after_step,
                "use_ema_warmup": self.use_ema_warmup, "inv_gamma": self.inv_gamma, "power": self.power, "shadow_params": self.shadow_params}

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        for k in ("decay", "min_decay", "optimization_step", "update_after_step", "use_ema_warmup", "inv_gamma", "power"):
            setattr(self, k, state_dict.get(k, getattr(self, k)))
        if not 0.0 <= self.decay <= 1.0:
            raise ValueError("Decay must be between 0 and 1")
        sp = state_dict.get("shadow_params", None)
        if sp is not None:
            if len(sp) != len(self.shadow_params) or not all(isinstance(t, torch.Tensor) for t in sp):
                raise ValueError("shadow_params must be a list of %d tensors" % len(self.shadow_params))
            for mine, theirs in zip(self.shadow_params, sp):
                mine.copy_(theirs)

    def save_pretrained(self, path):
        """the EMA weights as a model checkpoint whose config carries the EMA hyper-parameters (diffusers' layout: `unet_ema/`)"""
        if self.model_cls is None or self.model_config is None:
            raise ValueError("`save_pretrained` can only be used if `model_cls` and `model_config` were defined at __init__.")
        model = self.model_cls(**{k: v for k, v in dict(self.model_config).items() if not k.startswith("_")})
        sd = self.state_dict()
        sd.pop("shadow_params")
        model.register_to_config(**sd)
        with torch.no_grad():
            for s_param, param in zip(self.shadow_params, model.parameters()):
                param.copy_(s_param.to(param.device))
        model.save_pretrained(path)

    @classmethod
    def from_pretrained(cls, path, model_cls):
        """diffusers: `model_cls.load_config(path, return_unused_kwargs=True)` splits the EMA hyper-parameters off the model config; here the model loader
        ignores keys it does not know, and the EMA keys are read from the same config.json"""
        import json
        import os
        with open(os.path.join(path, model_cls.config_name)) as f:
            raw = json.load(f)
        ema_kwargs = {k: raw[k] for k in ("decay", "min_decay", "optimization_step", "update_after_step", "use_ema_warmup", "inv_gamma", "power") if k in raw}
        model = model_cls.from_pretrained(path)
        ema = cls(model.parameters(), model_cls=model_cls, model_config=model.config)
        ema.load_state_dict(ema_kwargs)
        return ema


def lr_lambda_for_world(lr_total_iter_length, lr_exp_warmup_steps, num_processes=None, final_ratio=0.01):
    """the `lr_func` of training/train.py:356: `accelerate` steps a prepared scheduler `num_processes` times per optimizer step, so the reference stretches BOTH
    lengths by the number of processes.  With `LambdaLR(FlatAdamW, ...)` stepped once per optimizer step (no accelerate wrapper) pass num_processes=1; under
    `accelerator.prepare(lr_scheduler)` pass the world size (default: torch.distributed's), exactly as the reference does."""
    if num_processes is None:
        num_processes = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    return IterExponential(total_iter_length=lr_total_iter_length * num_processes, final_ratio=final_ratio, warmup_steps=lr_exp_warmup_steps * num_processes)


def train_step(unet, vae, optimizer, batches, empty_encoding, modality="depth", lr_scale=1.0, gather_loss=False, train_batch_size=1):
    """One optimizer step over `batches` (a list of micro-batches = gradient accumulation, train.py:470,559-566): returns the
    mean micro-loss as a device tensor.  gather_loss=True: every micro-step's loss is averaged over the ranks as train.py:559 does for logging
    (`dist.gather_mean`: one small all-gather per micro-step, no host synchronisation).  train_batch_size: the CONFIGURED per-rank batch size the
    reference repeats the loss by (`args.train_batch_size`, a constant — never the size of the batch at hand: a ragged last batch would make the ranks'
    all-gather sizes differ; the repeat does not change the mean)."""
    from . import dist as D
    n = len(batches)
    total = None
    for i, batch in enumerate(batches):
        optimizer.sync_grads = i == n - 1
        loss = e2e_ft_loss(unet, vae, batch, empty_encoding, modality)
        (loss / n).backward()
        if gather_loss:
            loss = D.gather_mean(loss, train_batch_size)
        total = loss.detach() if total is None else total + loss.detach()
    optimizer.step(lr_scale=lr_scale)
    optimizer.zero_grad()
    return total / n


def synthetic_batch(batch, height, width, device, seed=0, dtype=torch.float32):
    """"Hypersim-val synthetic" micro-batch: uniform-noise rgb in [-1,1]; metric depth = tilted plane + 3 boxes in
    [0.5, 20] m, normalised to [-1,1] by its 2 % / 98 % quantiles (load.py:255-267); normals of that surface; 5 % invalid pixels."""
    g = torch.Generator(device=device).manual_seed(seed)
    rgb = (torch.randint(0, 256, (batch, 3, height, width), generator=g, device=device, dtype=torch.int32).float() / 255.0 * 2.0 - 1.0)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, height, device=device), torch.linspace(-1, 1, width, device=device), indexing="ij")
    depth = torch.empty((batch, 1, height, width), device=device)
    for b in range(batch):
        c = torch.rand(3, generator=g, device=device)
        d = 6.0 + 4.0 * (c[0] - 0.5) * xx + 4.0 * (c[1] - 0.5) * yy + 8.0 * c[2]
        for _ in range(3):
            r = torch.rand(5, generator=g, device=device)
            x0, y0 = r[0] * 1.4 - 1.0, r[1] * 1.4 - 1.0
            box = (xx > x0) & (xx < x0 + 0.2 + 0.5 * r[2]) & (yy > y0) & (yy < y0 + 0.2 + 0.5 * r[3])
            d = torch.where(box, 0.5 + 5.0 * r[4] + 0.0 * d, d)
        depth[b, 0] = d.clamp(0.5, 20.0)
    flat = depth.view(batch, -1)
    lo = torch.quantile(flat, 0.02, dim=1).view(batch, 1, 1, 1)
    hi = torch.quantile(flat, 0.98, dim=1).view(batch, 1, 1, 1)
    metric = (((depth - lo) / (hi - lo).clamp_min(1e-6) - 0.5) * 2.0).clamp(-1, 1)
    dzdx = torch.gradient(depth[:, 0], dim=2)[0]
    dzdy = torch.gradient(depth[:, 0], dim=1)[0]
    normals = torch.nn.functional.normalize(torch.stack([-dzdx * width / 2, -dzdy * height / 2, torch.ones_like(dzdx)], dim=1), dim=1)
    mask = torch.rand((batch, 1, height, width), generator=g, device=device) > 0.05
    return {"rgb": rgb.to(dtype), "metric": metric, "normals": normals, "val_mask": mask}
