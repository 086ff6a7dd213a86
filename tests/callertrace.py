"""Record / replay what the reference's CALLER code does with the `unet` and `vae` objects.  TEST INFRASTRUCTURE ONLY.

The drop-in boundary of this repository is "the product's `unet` / `vae` objects take the place of diffusers' in the reference's own
pipelines and training loop" (SURVEY.md §8b).  The reference's sources cannot travel to the GPU box and the product has no CPU path,
so the proof is split in two halves that meet in a committed fixture:

  record  (here, CPU; tests/golden/make_refwiring_golden.py): the reference's caller code — `MarigoldPipeline.single_infer`
          (Marigold/marigold/marigold_pipeline.py:372-538), `DepthNormalEstimationPipeline.single_infer`
          (GeoWizard/geowizard/models/geowizard_pipeline.py:252-401) and the training-step body (training/train.py:470-566) — is
          EXECUTED FROM ITS SOURCE over `Spy`-wrapped modules built from the reference's own vendored wiring.  Every attribute the
          caller reads and every call it makes (argument forms included: positional / keyword, 0-dim tensor timesteps,
          `return_dict=False` ...) is logged with its tensors; for the training step also the gradient that autograd sent into every
          call's output and got out of its inputs, and the UNet parameter gradients the reference's `accelerator.backward(loss)` left.
  replay  (GPU box; tests/test_reference_callers_gpu.py): each logged access / call is made on the PRODUCT object with the logged
          arguments; results (and vector-Jacobian products) must equal what the reference's modules returned.

A caller that only works because of something the trace does not show (an isinstance check, a private attribute) would fail in
`record`, because the Spy is not an nn.Module and exposes nothing but `__getattr__` and `__call__`.
"""
import torch

SCALARS = (int, float, bool, str, type(None))


def _pack(v):
    """tensors are stored detached; containers recursively; anything else by repr (never compared)"""
    if isinstance(v, torch.Tensor):
        return {"__tensor__": v.detach().clone()}
    if isinstance(v, SCALARS):
        return v
    if isinstance(v, (list, tuple)):
        return {"__seq__": [_pack(x) for x in v], "tuple": isinstance(v, tuple)}
    if hasattr(v, "sample") and isinstance(getattr(v, "sample"), torch.Tensor):       # UNet2DConditionOutput
        return {"__output__": {"sample": _pack(v.sample)}}
    if hasattr(v, "image_embeds"):
        return {"__output__": {"image_embeds": _pack(v.image_embeds)}}
    return {"__repr__": type(v).__name__}


class Spy:
    """Transparent proxy: logs attribute reads and calls under a dotted path ("unet", "vae.encoder", "vae.config.scaling_factor")."""

    def __init__(self, target, path, log, vjp=False):
        object.__setattr__(self, "_t", target)
        object.__setattr__(self, "_p", path)
        object.__setattr__(self, "_log", log)
        object.__setattr__(self, "_vjp", vjp)

    def __getattr__(self, name):
        val = getattr(self._t, name)
        path = self._p + "." + name
        if isinstance(val, torch.Tensor) or isinstance(val, SCALARS):
            self._log.append({"kind": "getattr", "path": path, "value": _pack(val)})
            return val
        if callable(val) and not isinstance(val, torch.nn.Module) and not hasattr(val, "keys"):
            # bound method (parameters(), train(), to(), requires_grad_() ...): log the call, hand the real result back
            def method(*a, **k):
                self._log.append({"kind": "method", "path": path, "args": _pack(list(a)), "kwargs": {kk: _pack(vv) for kk, vv in k.items()}})
                return val(*a, **k)
            return method
        return Spy(val, path, self._log, self._vjp)

    def __getitem__(self, key):                 # config['in_channels']
        val = self._t[key]
        self._log.append({"kind": "getitem", "path": self._p, "key": key, "value": _pack(val)})
        return val

    def __call__(self, *args, **kwargs):
        out = self._t(*args, **kwargs)
        ev = {"kind": "call", "path": self._p, "args": _pack(list(args)), "kwargs": {k: _pack(v) for k, v in kwargs.items()}, "out": _pack(out)}
        self._log.append(ev)
        if self._vjp and torch.is_grad_enabled():
            t_out = out if isinstance(out, torch.Tensor) else (out[0] if isinstance(out, (tuple, list)) else getattr(out, "sample", None))
            if isinstance(t_out, torch.Tensor) and t_out.requires_grad:
                t_out.register_hook(lambda g, ev=ev: ev.__setitem__("grad_out", g.detach().clone()))
                ev["grad_args"] = {}
                for i, a in enumerate(args):
                    if isinstance(a, torch.Tensor) and a.requires_grad:
                        a.register_hook(lambda g, ev=ev, i=i: ev["grad_args"].__setitem__(i, g.detach().clone()))
        return out


# ---- replay ------------------------------------------------------------------------------------------------------------------------
def unpack(v, device=None, dtype=None, requires_grad=False):
    if isinstance(v, dict):
        if "__tensor__" in v:
            t = v["__tensor__"]
            if device is not None:
                t = t.to(device)
            if dtype is not None and t.is_floating_point():
                t = t.to(dtype)
            if requires_grad and t.is_floating_point():
                t = t.clone().requires_grad_(True)
            return t
        if "__seq__" in v:
            s = [unpack(x, device, dtype, requires_grad) for x in v["__seq__"]]
            return tuple(s) if v["tuple"] else s
        if "__output__" in v:
            return {k: unpack(x, device, dtype) for k, x in v["__output__"].items()}
        return v
    return v


def resolve(roots, path):
    head, *rest = path.split(".")
    obj = roots[head]
    for name in rest:
        obj = getattr(obj, name)
    return obj


def first_tensor(out):
    """what the caller took from a call's result: the tensor itself, `[0]` of a tuple, or `.sample`"""
    if isinstance(out, torch.Tensor):
        return out
    if isinstance(out, (tuple, list)):
        return out[0]
    if isinstance(out, dict):
        return out.get("sample", next(iter(out.values())))
    return out.sample if hasattr(out, "sample") else out.image_embeds
