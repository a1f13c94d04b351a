"""Dice + CE compound loss, one fused HIP pass; drop-in for the reference ``DiceAndCELoss``
(`src/losses/compound_losses.py:17-65`)."""
from __future__ import annotations

from typing import Callable

import torch
from torch import nn

from mia_hip import ops

from .ce_loss import RobustCrossEntropyLoss, hip_cross_entropy
from .dice_loss import DiceLoss


class DiceAndCELoss(nn.Module):
    def __init__(self, dice_loss: Callable = DiceLoss, dice_kwargs: dict = {}, ce_loss: Callable = RobustCrossEntropyLoss,
                 ce_kwargs: dict = {}, default_dice_weight: float = 1.0, default_ce_weight: float = 1.0):
        super().__init__()
        self.dice_loss = dice_loss(**dice_kwargs)
        self.ce_loss = ce_loss(**ce_kwargs)
        self.default_dice_weight = default_dice_weight
        self.default_ce_weight = default_ce_weight

    def _fusable(self):
        return isinstance(self.dice_loss, DiceLoss) and isinstance(self.ce_loss, nn.CrossEntropyLoss)

    def forward(self, outputs: torch.Tensor, targets: torch.Tensor, dice_weight: float | None = None,
                ce_weight: float | None = None):
        if not dice_weight:  # reference quirk: 0.0 / None -> default (compound_losses.py:40-44)
            dice_weight = self.default_dice_weight
        if not ce_weight:
            ce_weight = self.default_ce_weight
        if self._fusable():
            d = self.dice_loss
            d._check(outputs, targets)
            if getattr(self.ce_loss, "weight", None) is not None or self.ce_loss.label_smoothing != 0.0 or \
                    self.ce_loss.reduction != "mean":
                raise NotImplementedError("HIP cross-entropy implements the al_train configuration only")
            return ops.DiceCEFn.apply(outputs, targets, d._flags(), float(d.smooth), float(dice_weight), float(ce_weight), 0)
        return ce_weight * self.get_ce_loss(outputs, targets) + dice_weight * self.dice_loss(outputs, targets)

    def get_dice_loss(self, outputs: torch.Tensor, targets: torch.Tensor):
        return self.dice_loss(outputs, targets)

    def get_ce_loss(self, outputs: torch.Tensor, targets: torch.Tensor):
        if isinstance(self.ce_loss, nn.CrossEntropyLoss):
            if targets.ndim == outputs.ndim and targets.shape[1] == 1 and outputs.shape[1] > 1:
                targets = targets[:, 0]
            return hip_cross_entropy(self.ce_loss, outputs, targets)
        return self.ce_loss(outputs, targets)
