"""Poly LR with linear warm-up (host scalar math); drop-in for the reference ``PolyLRScheduler``
(`src/scheduler/lr_scheduler.py:6-55`).  Works with any object exposing ``param_groups``
(torch optimizers and training.FlatOptimizer)."""
from __future__ import annotations

from typing import Optional

import torch


def poly_lr(step: int, initial_lr: float, max_steps: int, warmup_steps: int, exponent: float = 0.9, interval: int = 1) -> float:
    idx = step // interval
    warm = warmup_steps // interval
    total = max_steps // interval
    if idx < warm:
        return initial_lr * (idx + 1) / warm
    idx -= warm
    return initial_lr * (1.0 - idx / (total - warm)) ** exponent


class PolyLRScheduler:
    def __init__(self, optimizer, initial_lr: float, max_steps: int, warmup_steps: int, exponent: float = 0.9,
                 current_step: int | None = None, interval: int = 1):
        self.optimizer = optimizer
        self.initial_lr = initial_lr
        self.max_steps = max_steps
        self.warmup_steps = warmup_steps
        self.exponent = exponent
        self.interval = interval
        self.ctr = 0
        self.last_epoch = current_step if current_step is not None else -1
        self._last_lr = [g["lr"] for g in optimizer.param_groups]

    def step(self, epoch: Optional[int] = None):
        if epoch is None or epoch == -1:
            epoch = self.ctr
            self.ctr += 1
        new_lr = poly_lr(epoch, self.initial_lr, self.max_steps, self.warmup_steps, self.exponent, self.interval)
        for g in self.optimizer.param_groups:
            if isinstance(g["lr"], torch.Tensor):
                g["lr"].fill_(new_lr)
            else:
                g["lr"] = new_lr
        self._last_lr = [g["lr"] for g in self.optimizer.param_groups]

    def get_last_lr(self):
        return self._last_lr
