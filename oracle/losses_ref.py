"""CPU restatement of the reference Dice / CE / Dice+CE losses (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/src/losses/dice_loss.py:7-76 (DiceLoss),
src/losses/ce_loss.py:6-16 (RobustCrossEntropyLoss),
src/losses/compound_losses.py:17-65 (DiceAndCELoss).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F


def one_hot(labels: torch.Tensor, k1: int) -> torch.Tensor:
    """dice_loss.py:25-30 -- [B,H,W] int -> [B,K1,H,W] float."""
    return F.one_hot(labels.long(), k1).permute(0, 3, 1, 2).to(torch.float32)


def dice_loss(outputs: torch.Tensor, targets: torch.Tensor, num_classes: int, smooth: float = 1e-5,
              do_bg: bool = False, softmax: bool = True, batch: bool = False,
              squared: bool = False) -> torch.Tensor:
    """dice_loss.py:32-76.  ``num_classes`` is the constructor argument (foreground
    classes); the loss works on ``num_classes + 1`` channels (dice_loss.py:18)."""
    k1 = num_classes + 1
    if softmax:
        outputs = torch.softmax(outputs, dim=1)
    if outputs.shape != targets.shape:
        targets = one_hot(targets, k1)
    if not do_bg:
        outputs, targets = outputs[:, 1:], targets[:, 1:]
    assert outputs.shape == targets.shape
    axes = tuple(range(2, outputs.ndim))
    inter = (outputs * targets).sum(axes)
    if squared:
        s_in, s_t = (outputs ** 2).sum(axes), (targets ** 2).sum(axes)
    else:
        s_in, s_t = outputs.sum(axes), targets.sum(axes)
    if batch:
        inter, s_in, s_t = inter.mean(0), s_in.mean(0), s_t.mean(0)
    dice = 1 - (2 * inter + smooth) / (s_in + s_t + smooth)
    return dice.mean()


def ce_loss(outputs: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """ce_loss.py:12-16 / torch.nn.CrossEntropyLoss(): mean over all pixels.  A target with the logits' shape is a
    class-probability target (torch.nn.CrossEntropyLoss semantics; reached through DiceAndCELoss with dense targets)."""
    if targets.shape == outputs.shape and outputs.shape[1] > 1:
        return F.cross_entropy(outputs, targets.float())
    if targets.ndim == outputs.ndim:
        assert targets.shape[1] == 1
        targets = targets[:, 0]
    return F.cross_entropy(outputs, targets.long())


def dice_and_ce(outputs: torch.Tensor, targets: torch.Tensor, num_classes: int,
                dice_weight: Optional[float] = None, ce_weight: Optional[float] = None,
                default_dice_weight: float = 1.0, default_ce_weight: float = 1.0,
                smooth: float = 1e-5, do_bg: bool = True, softmax: bool = True, batch: bool = False,
                squared: bool = False) -> torch.Tensor:
    """compound_losses.py:33-49, with al_train's dice kwargs as defaults
    (al_trainer.py:786-793).  Note the ``if not weight`` quirk: 0.0/None -> default."""
    if not dice_weight:
        dice_weight = default_dice_weight
    if not ce_weight:
        ce_weight = default_ce_weight
    l_ce = ce_loss(outputs, targets)
    l_dice = dice_loss(outputs, targets, num_classes, smooth, do_bg, softmax, batch, squared)
    return ce_weight * l_ce + dice_weight * l_dice


def hard_dice(pred: torch.Tensor, gt: torch.Tensor) -> float:
    """medpy.metric.dc closed form used by al_trainer.py:1539-1556:
    2|A&B| / (|A|+|B|); 0 when the prediction is empty (al_trainer.py:1548)."""
    pred, gt = pred.bool(), gt.bool()
    if pred.sum() == 0:
        return 0.0
    inter = (pred & gt).sum().item()
    denom = pred.sum().item() + gt.sum().item()
    return 2.0 * inter / denom if denom > 0 else 0.0
