"""Validation reductions on the GPU: label maps + per-class hard Dice in one pass over the logits.

Replaces, for the Dice part, the CPU loop of ``ALTrainer.valid_slices`` / ``calculate_metric_percase``
(reference `src/training/al_trainer.py:1428-1431`, `:1463-1472`, `:1539-1556`: ``softmax -> argmax -> .cpu().numpy()``
then ``medpy.metric.dc`` per image and class).  Hausdorff / ASD stay out of scope (SimpleITK / medpy host code)."""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from mia_hip import call, lib
from mia_hip.ops import _c_i64, _need_dev, _p, _pix_strides, _stream


def predict_and_dice(logits: torch.Tensor, labels: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]:
    """logits [B,K1,H,W] fp32 (any pixel-collapsible strides) -> (pred [B,H,W] int64, dice [B,K1], counts [B,K1,3]).

    ``dice[b, k]`` = 2|P&G| / (|P|+|G|) for class k (``pred == k`` vs ``label == k``), 0 where the prediction is empty
    -- `calculate_metric_percase` semantics; the trainer's "all foreground" metric is
    ``2*sum_k>0 I / (sum_k>0 P + sum_k>0 G)`` from ``counts``."""
    _need_dev(logits, labels)
    if logits.dtype != torch.float32:
        logits = logits.float()
    st = _pix_strides(logits)
    if st is None:
        logits = logits.contiguous()
        st = _pix_strides(logits)
    b, k1, h, w = logits.shape
    hw = h * w
    slabs = max(1, min(128, hw // 2048))
    dev = logits.device
    pred = torch.empty((b, h, w), device=dev, dtype=torch.long)
    ws = counts = dice = None
    lab = None
    if labels is not None:
        lab = labels.reshape(b, h, w).long().contiguous()
        ws = torch.empty(lib().mia_argmax_dice_workspace(b, k1, slabs), device=dev, dtype=torch.float32)
        counts = torch.empty((b, k1, 3), device=dev, dtype=torch.float32)
        dice = torch.empty((b, k1), device=dev, dtype=torch.float32)
    call("mia_argmax_dice", _p(logits), _p(lab), _p(pred), b, _c_i64(hw), k1, _c_i64(st[0]), _c_i64(st[1]), _c_i64(st[2]), slabs,
         _p(ws), _p(counts), _p(dice), _stream())
    return pred, dice, counts
