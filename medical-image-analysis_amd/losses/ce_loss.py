"""Cross-entropy on the fused HIP Dice/CE kernel (reference `src/losses/ce_loss.py:6-16`)."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from mia_hip import ops


class RobustCrossEntropyLoss(nn.CrossEntropyLoss):
    """Accepts a [B,1,H,W] (possibly float) target like the reference; mean over all pixels.
    Class weights / ignore_index / label smoothing are not on the al_train path and are rejected."""

    def forward(self, input: Tensor, target: Tensor) -> Tensor:
        if target.ndim == input.ndim:
            assert target.shape[1] == 1
            target = target[:, 0]
        return hip_cross_entropy(self, input, target)


def hip_cross_entropy(module, input: Tensor, target: Tensor) -> Tensor:
    if getattr(module, "weight", None) is not None or getattr(module, "label_smoothing", 0.0) != 0.0 or \
            getattr(module, "reduction", "mean") != "mean":
        raise NotImplementedError("HIP cross-entropy implements the al_train configuration: no class weights, "
                                  "no label smoothing, reduction='mean'")
    ign = getattr(module, "ignore_index", -100)
    if 0 <= ign < input.shape[1]:
        raise NotImplementedError("HIP cross-entropy has no ignore_index: every label must be a class in [0, K1)")
    # the default ignore_index (-100) is not honoured either: such a label is out of range -> NaN loss, ops.check_labels() raises
    if target.shape != input.shape:
        target = target.long()
    return ops.DiceCEFn.apply(input, target, ops.loss_flags(True, True, False, False), 1e-5, 0.0, 1.0, 0)
