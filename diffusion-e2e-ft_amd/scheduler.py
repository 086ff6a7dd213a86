"""DDIM scheduler surface used by the pipelines (diffusers DDIMScheduler semantics; scaled_linear betas,
v_prediction, trailing spacing — Marigold/run.py:157-162, training/train.py:509-518,613-617)."""
import numpy as np
import torch

from .unet import Config


class SchedulerOutput:
    def __init__(self, prev_sample, pred_original_sample):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample


class DDIMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 prediction_type="v_prediction", timestep_spacing="trailing", steps_offset=1, clip_sample=False,
                 set_alpha_to_one=False, thresholding=False):
        if beta_schedule != "scaled_linear":
            raise NotImplementedError(beta_schedule)
        self.config = Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                             beta_schedule=beta_schedule, prediction_type=prediction_type, timestep_spacing=timestep_spacing,
                             steps_offset=steps_offset, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
                             thresholding=thresholding)
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    config_name = "scheduler_config.json"

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **overrides):
        """DDIMScheduler.from_pretrained(ckpt, timestep_spacing=..., subfolder="scheduler") (Marigold/run.py:272): diffusers'
        scheduler_config.json, keyword overrides win, unknown keys (trained_betas, rescale_betas_zero_snr, ...) are ignored"""
        import inspect
        import json
        import os
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, cls.config_name)) as f:
            cfg = json.load(f)
        cfg.update(overrides)
        known = set(inspect.signature(cls.__init__).parameters) - {"self"}
        return cls(**{k: v for k, v in cfg.items() if k in known})

    def save_pretrained(self, save_directory, **kw):
        import json
        import os
        os.makedirs(save_directory, exist_ok=True)
        cfg = dict(self.config)
        cfg["_class_name"] = type(self).__name__
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2)

    def set_timesteps(self, num_inference_steps, device=None):
        n, T = num_inference_steps, self.config.num_train_timesteps
        self.num_inference_steps = n
        sp = self.config.timestep_spacing
        if sp == "trailing":
            ts = np.round(np.arange(T, 0, -T / n)) - 1
        elif sp == "leading":
            ts = (np.arange(0, n) * (T // n)).round()[::-1].copy() + self.config.steps_offset
        elif sp == "linspace":
            ts = np.linspace(0, T - 1, n).round()[::-1].copy()
        else:
            raise ValueError(sp)
        self.timesteps_host = [int(v) for v in ts]           # host copy: reading a timestep must not synchronise with the device
        key = (tuple(self.timesteps_host), str(device))
        if getattr(self, "_ts_key", None) != key:            # same schedule as last call: keep the device tensor (no H2D per image)
            self.timesteps = torch.from_numpy(ts.astype(np.int64)).to(device)
            self._ts_key = key

    def x0_coefficients(self, t):
        """(sqrt(abar_t), sqrt(1 - abar_t)) as python floats computed in fp32 like the reference (train.py:509-512)."""
        ac = self.alphas_cumprod[int(t)]
        return float(ac ** 0.5), float((1 - ac) ** 0.5)

    def zero_latent_x0_scale(self, t):
        """c with x0 = c * model_output when x_t = 0 (the E2E-FT recipe, training/train.py:480-518): the scheduler's own
        `prediction_type` decides — v_prediction -sqrt(1 - abar_t), epsilon -sqrt(1 - abar_t) / sqrt(abar_t), sample 1 — as in step().
        clip_sample / thresholding configurations are refused rather than silently ignored by the fused single-step paths."""
        if self.config.clip_sample or self.config.thresholding:
            raise NotImplementedError("the fused single-step path does not clip / threshold x0 (scheduler config clip_sample=%s, thresholding=%s)"
                                      % (self.config.clip_sample, self.config.thresholding))
        sa, sb = self.x0_coefficients(t)
        pt = self.config.prediction_type
        if pt == "v_prediction":
            return -sb
        if pt == "epsilon":
            return -sb / sa
        if pt == "sample":
            return 1.0
        raise ValueError("Unknown prediction type %s" % pt)

    def step(self, model_output, timestep, sample, eta=0.0, **kw):
        """DDIM step for epsilon / sample / v_prediction with eta = 0 (elementwise torch ops on tiny latents; the
        pipelines' fused path uses ops.copy_scale instead)."""
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t].item()
        a_prev = (self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod).item()
        b_t = 1 - a_t
        pt = self.config.prediction_type
        if pt == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif pt == "sample":
            x0 = model_output
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        elif pt == "v_prediction":
            x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
            eps = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
        else:
            raise ValueError(pt)
        if self.config.clip_sample or self.config.thresholding:
            raise NotImplementedError("clip_sample / thresholding are not implemented (SD-v2 / Marigold / GeoWizard schedulers set neither)")
        prev = a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps
        return SchedulerOutput(prev, x0)


class DDPMScheduler(DDIMScheduler):
    """The name the training script loads, reads and saves (training/train.py:25,292,461,480,511-524,613-617): `DDPMScheduler.from_pretrained(ckpt,
    subfolder="scheduler"[, timestep_spacing="trailing"])`, `.alphas_cumprod`, `.config.num_train_timesteps / prediction_type / thresholding /
    clip_sample`, and at the end of training `StableDiffusionPipeline(..., scheduler=DDPMScheduler.from_pretrained(..., timestep_spacing="trailing"))
    .save_pretrained(out)` which writes scheduler/scheduler_config.json.  The E2E-FT step never calls the scheduler's stochastic `step()`: it only
    needs the noise schedule (identical to DDIM's: scaled_linear betas, diffusers scheduling_ddpm.py) and the config, so this is the same object
    under the reference's class name; `save_pretrained` records `_class_name: DDPMScheduler` as diffusers would.  SD-v2 ships `clip_sample: false`;
    a checkpoint that sets it (DDPM's diffusers default is true) is refused by the fused single-step path exactly like DDIM's (zero_latent_x0_scale)."""

    def step(self, *a, **kw):
        raise NotImplementedError("ancestral DDPM sampling is not on the E2E-FT path (training/train.py reads the schedule only; inference uses DDIMScheduler)")
