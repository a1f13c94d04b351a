"""CPU restatement of ``ALTrainer.train_step`` (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/src/training/al_trainer.py:1350-1399 (train_step),
:737-780 (_setup_optimizer), :782-800 (_setup_loss) and
src/scheduler/lr_scheduler.py:31-55 (PolyLRScheduler.step).  Also used by
``bench.py`` as the ``cpu_baseline`` ("port") leg.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import losses_ref, unet_ref


def poly_lr(step: int, initial_lr: float, max_steps: int, warmup_steps: int, exponent: float = 0.9,
            interval: int = 1) -> float:
    """lr_scheduler.py:31-48."""
    idx = step // interval
    w = warmup_steps // interval
    m = max_steps // interval
    if idx < w:
        return initial_lr * (idx + 1) / w
    idx -= w
    return initial_lr * (1.0 - idx / (m - w)) ** exponent


def trainable(p: unet_ref.Params):
    return {k: v for k, v in p.items() if v.is_floating_point() and "running_" not in k}


def make_optimizer(p: unet_ref.Params, name: str = "adamw", **kwargs) -> torch.optim.Optimizer:
    """al_trainer.py:744-761 -- no lr at construction (torch default until the
    scheduler overwrites it)."""
    params = [v.requires_grad_(True) for v in trainable(p).values()]
    if name == "adam":
        return torch.optim.Adam(params, betas=(0.9, 0.999), **kwargs)
    if name == "adamw":
        return torch.optim.AdamW(params, betas=(0.9, 0.999), **kwargs)
    if name == "sgd":
        return torch.optim.SGD(params, momentum=0.9, **kwargs)
    raise ValueError(f'Optimizer "{name}" not supported')


def train_step(p: unet_ref.Params, opt: torch.optim.Optimizer, image: torch.Tensor, label: torch.Tensor,
               num_classes: int, normalization: str = "instance", lr: Optional[float] = None,
               max_grad_norm: float = 10.0, drop_masks: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """One iteration of al_trainer.py:1350-1381: set lr, forward, Dice+CE, zero_grad,
    backward, clip_grad_norm_(10), optimizer.step()."""
    if lr is not None:
        for g in opt.param_groups:
            g["lr"] = lr
    out = unet_ref.unet_forward(p, image.float(), normalization, True, drop_masks=drop_masks)
    loss = losses_ref.dice_and_ce(out, label.long(), num_classes)
    opt.zero_grad()
    loss.backward()
    params = [q for g in opt.param_groups for q in g["params"]]
    gn = torch.nn.utils.clip_grad_norm_(params, max_norm=max_grad_norm)
    opt.step()
    return {"loss": loss.detach(), "logits": out.detach(), "grad_norm": gn.detach()}
