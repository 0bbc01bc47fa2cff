"""Learning-rate multipliers of ``diffusers.optimization.get_scheduler`` (diffusers==0.19.3, not vendored by the
reference) as the reference uses it at exp-1-debias-gender/1-main-debias.py:1639-1646, 2023.

The reference passes ``num_warmup_steps * P`` and ``num_training_steps * P`` and accelerate's prepared scheduler
advances P ticks per call, so in units of training steps the multiplier is ``lr_lambda(name, n, warmup, total)``;
``lr_scheduler.step()`` runs every step, also when a non-finite gradient skipped the optimiser (:2018-2023).
Only ``cosine_with_restarts`` receives ``num_cycles`` and only ``polynomial`` receives ``power``.
"""
import math

SCHEDULES = ("linear", "cosine", "cosine_with_restarts", "polynomial", "constant", "constant_with_warmup")


def lr_lambda(name, step, num_warmup_steps=0, num_training_steps=1, num_cycles=1, power=1.0, lr_init=1.0):
    w, T = num_warmup_steps, num_training_steps
    if name == "constant":
        return 1.0
    if name not in SCHEDULES:
        raise ValueError(f"unknown lr_scheduler {name!r}; expected one of {SCHEDULES}")
    if step < w:
        return float(step) / float(max(1, w)) if name != "constant_with_warmup" else float(step) / float(max(1.0, w))
    if name == "constant_with_warmup":
        return 1.0
    if name == "linear":
        return max(0.0, float(T - step) / float(max(1, T - w)))
    progress = float(step - w) / float(max(1, T - w))
    if name == "cosine":
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * progress)))
    if name == "cosine_with_restarts":
        if progress >= 1.0:
            return 0.0
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(num_cycles) * progress) % 1.0))))
    # polynomial: decays to lr_end = 1e-7 (absolute), expressed as a multiplier of lr_init
    lr_end = 1e-7
    if step > T:
        return lr_end / lr_init
    pct_remaining = 1 - (step - w) / (T - w)
    return ((lr_init - lr_end) * pct_remaining ** power + lr_end) / lr_init
