"""CPU restatement of the reference's training hooks.  TEST INFRASTRUCTURE ONLY.
PINNED: tests/test_oracle_pins.py compares every function here with the reference's own module imported from
/root/reference (when present) and with fixtures generated from it (tests/golden/make_golden.py)."""
import math

import torch


def compute_scale_and_shift_masked_ref(prediction, target, mask):
    """training/util/loss.py:31-47 — closed-form masked least squares, det > 0 guard."""
    m = mask.to(prediction.dtype)
    a_00 = torch.sum(m * prediction * prediction, (1, 2))
    a_01 = torch.sum(m * prediction, (1, 2))
    a_11 = torch.sum(m, (1, 2))
    b_0 = torch.sum(m * prediction * target, (1, 2))
    b_1 = torch.sum(m * target, (1, 2))
    x_0 = torch.zeros_like(b_0)
    x_1 = torch.zeros_like(b_1)
    det = a_00 * a_11 - a_01 * a_01
    valid = det > 0
    x_0[valid] = (a_11[valid] * b_0[valid] - a_01[valid] * b_1[valid]) / det[valid]
    x_1[valid] = (-a_01[valid] * b_0[valid] + a_00[valid] * b_1[valid]) / det[valid]
    return x_0, x_1


def ssi_loss_ref(prediction, target, mask):
    """training/util/loss.py:13-29 — ScaleAndShiftInvariantLoss.forward (fp32)."""
    if mask.ndim == 4:
        mask = mask.squeeze(1)
    prediction, target = prediction.squeeze(1).float(), target.squeeze(1).float()
    scale, shift = compute_scale_and_shift_masked_ref(prediction, target, mask)
    scaled = scale.view(-1, 1, 1) * prediction + shift.view(-1, 1, 1)
    return torch.nn.functional.l1_loss(scaled[mask], target[mask])


def angular_loss_ref(prediction, target, mask):
    """training/util/loss.py:51-67 — AngularLoss.forward (fp32, mask channel 0)."""
    prediction, target = prediction.float(), target.float()
    m = mask[:, 0, :, :]
    dot = torch.clamp(torch.sum(prediction * target, dim=1), -1.0, 1.0)
    return torch.acos(dot)[m].mean()


def iter_exponential_ref(n_iter, total_iter_length, final_ratio, warmup_steps=0):
    """training/util/lr_scheduler.py:10-36 — IterExponential.__call__."""
    if n_iter < warmup_steps:
        return 1.0 * n_iter / warmup_steps
    if n_iter >= total_iter_length:
        return final_ratio
    return math.exp((n_iter - warmup_steps) / (total_iter_length - warmup_steps) * math.log(final_ratio))


def replace_conv_in_ref(weight, bias, repeat=2):
    """training/util/unet_prep.py:6-20 — returns the new (weight, bias) of conv_in: Cin repeated, both halved."""
    return weight.repeat(1, repeat, 1, 1) / repeat, bias / repeat
