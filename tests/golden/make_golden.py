"""Generate the golden fixtures (run here, on CPU, from the repo root: `python tests/golden/make_golden.py`).

  model_golden.pt  — outputs of the CPU oracle (oracle/) on seeded synthetic weights/inputs for the tiny configs.
                     PARITY UNPINNED at the diffusers boundary: these freeze the oracle, they do not
                     come from the reference implementation (diffusers is not installable here, no checkpoints exist).
  train_golden.pt  — loss, estimate and UNet gradients of one E2E-FT micro-step (depth / normals) through torch autograd over
                     the CPU oracle (same pinning caveat as model_golden.pt).
  hooks_golden.pt  — outputs of the REFERENCE'S OWN modules imported from /root/reference/training/util
                     (loss.py, lr_scheduler.py, unet_prep.py) on seeded inputs: these pin oracle/losses_ref.py and the
                     HIP loss kernels to the reference.
Inputs are regenerated from seeds by the tests (oracle.synth), only outputs are stored."""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import config, unet_ref, vae_ref, pipeline_ref, synth  # noqa: E402
import golden_cases as gc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _import(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    torch.set_num_threads(8)
    out = {}
    with torch.no_grad():
        for name, fn in gc.MODEL_CASES.items():
            out[name] = fn()
            print(name, {k: tuple(v.shape) for k, v in out[name].items()})
    torch.save(out, os.path.join(HERE, "model_golden.pt"))
    train = {m: gc.train_grads(m) for m in ("depth", "normals")}
    train["geowizard"] = gc.geo_train_grads()
    torch.save(train, os.path.join(HERE, "train_golden.pt"))
    print("train:", {m: (float(v["loss"]), len(v["grads"]), len(v["grad_norms"])) for m, v in train.items()})

    ref_dir = "/root/reference/training/util"
    if os.path.isdir(ref_dir):
        loss = _import(os.path.join(ref_dir, "loss.py"), "ref_loss")
        lr = _import(os.path.join(ref_dir, "lr_scheduler.py"), "ref_lr")
        prep = _import(os.path.join(ref_dir, "unet_prep.py"), "ref_prep")
        hooks = {}
        pred, tgt, mask = gc.ssi_inputs()
        hooks["ssi_loss"] = loss.ScaleAndShiftInvariantLoss()(pred, tgt, mask)
        pg = pred.clone().requires_grad_(True)
        loss.ScaleAndShiftInvariantLoss()(pg, tgt, mask).backward()
        hooks["ssi_dpred"] = pg.grad.clone()
        s, t = loss.compute_scale_and_shift_masked(pred.squeeze(1), tgt.squeeze(1), mask.squeeze(1))
        hooks["ssi_scale"], hooks["ssi_shift"] = s, t
        n, nt, m3 = gc.angular_inputs()
        hooks["angular_loss"] = loss.AngularLoss()(n, nt, m3)
        ng = n.clone().requires_grad_(True)
        loss.AngularLoss()(ng, nt, m3).backward()
        hooks["angular_dpred"] = ng.grad.clone()
        sched = lr.IterExponential(total_iter_length=20000, final_ratio=0.01, warmup_steps=100)
        hooks["lr_iters"] = torch.tensor(gc.LR_ITERS)
        hooks["lr_values"] = torch.tensor([float(sched(i)) for i in gc.LR_ITERS], dtype=torch.float64)

        class _U:  # minimal stand-in exposing what replace_unet_conv_in touches
            pass
        u = _U()
        g = torch.Generator().manual_seed(77)
        u.conv_in = torch.nn.Conv2d(4, 32, 3, 1, 1)
        with torch.no_grad():
            u.conv_in.weight.copy_(torch.randn(32, 4, 3, 3, generator=g))
            u.conv_in.bias.copy_(torch.randn(32, generator=g))
        u.config = {"in_channels": 4}
        w0, b0 = u.conv_in.weight.detach().clone(), u.conv_in.bias.detach().clone()
        prep.replace_unet_conv_in(u, repeat=2)
        hooks["conv_in_w0"], hooks["conv_in_b0"] = w0, b0
        hooks["conv_in_w"], hooks["conv_in_b"] = u.conv_in.weight.detach().clone(), u.conv_in.bias.detach().clone()
        hooks["conv_in_cfg"] = torch.tensor(u.config["in_channels"])
        torch.save(hooks, os.path.join(HERE, "hooks_golden.pt"))
        print("hooks:", {k: (tuple(v.shape) if hasattr(v, "shape") else v) for k, v in hooks.items()})
    else:
        print("reference not present: hooks_golden.pt not regenerated")


if __name__ == "__main__":
    main()
