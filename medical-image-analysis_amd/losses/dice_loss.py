"""Dice loss on the fused HIP Dice/CE kernel; drop-in for the reference ``DiceLoss``
(`src/losses/dice_loss.py:7-76`): same constructor, ``num_classes`` means FOREGROUND classes
(``self.num_classes = num_classes + 1``), returns a 0-dim tensor with autograd."""
from __future__ import annotations

import torch
from torch import nn

from mia_hip import ops


class DiceLoss(nn.Module):
    def __init__(self, num_classes: int, smooth: float = 1e-5, do_bg: bool = False, softmax: bool = True,
                 batch: bool = False, squared: bool = False):
        super().__init__()
        self.num_classes = num_classes + 1  # include background (reference dice_loss.py:18)
        self.smooth = smooth
        self.do_bg = do_bg
        self.softmax = softmax
        self.batch = batch
        self.squared = squared

    def _flags(self):
        return ops.loss_flags(self.softmax, self.do_bg, self.batch, self.squared)

    def _check(self, outputs, targets):
        if outputs.ndim != 4:
            raise NotImplementedError("MI355X DiceLoss implements 2-D inputs [B, K, H, W]")
        if targets.shape == outputs.shape and outputs.shape[1] > 1:
            return  # dense (one-hot / soft) target, used as is (reference dice_loss.py:40-41)
        assert outputs.shape[1] == self.num_classes, "inputs {} & num_classes+1 {} do not match".format(
            tuple(outputs.shape), self.num_classes)
        assert targets.numel() == outputs.shape[0] * outputs.shape[2] * outputs.shape[3], \
            "inputs {} & target {} shape do not match".format(outputs.size(), targets.size())

    def forward(self, outputs: torch.Tensor, targets: torch.Tensor):
        self._check(outputs, targets)
        return ops.DiceCEFn.apply(outputs, targets, self._flags(), float(self.smooth), 1.0, 0.0, 0)
