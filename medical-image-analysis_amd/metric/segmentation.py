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


def label_dice(pred: torch.Tensor, labels: torch.Tensor, k1: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-(image, class) hard Dice of two LABEL MAPS [B,H,W] (int64) -> (dice [B,K1], counts [B,K1,3] = |P&G|, |P|, |G|):
    `calculate_metric_percase` (al_trainer.py:1539-1556) on `pred == k` vs `label == k`, all classes in one pass."""
    _need_dev(pred, labels)
    b, h, w = pred.shape
    hw = h * w
    slabs = max(1, min(128, hw // 2048))
    dev = pred.device
    p = pred.long().contiguous()
    lab = labels.reshape(b, h, w).long().contiguous()
    ws = torch.empty(lib().mia_argmax_dice_workspace(b, k1, slabs), device=dev, dtype=torch.float32)
    counts = torch.empty((b, k1, 3), device=dev, dtype=torch.float32)
    dice = torch.empty((b, k1), device=dev, dtype=torch.float32)
    call("mia_argmax_dice", None, _p(lab), _p(p), b, _c_i64(hw), k1, _c_i64(0), _c_i64(0), _c_i64(0), slabs, _p(ws), _p(counts),
         _p(dice), _stream())
    return dice, counts


@torch.no_grad()
def valid_slices(model, processor, image_batch: torch.Tensor, label_batch: torch.Tensor, num_classes: int, loss_fn=None,
                 do_denoise: bool = False):
    """One validation step of `ALTrainer.valid_slices` (al_trainer.py:1415-1474) chained on the GPU:
    `processor.preprocess` (bilinear resize to the model size, unet_processor.py:35-47) -> eval forward ->
    `softmax(1).argmax(1)` -> (loss on labels nearest-resized to the output size, :1433-1449) -> `processor.postprocess`
    (nearest resize back to the label size, unet_processor.py:49-70) -> hard Dice at the ORIGINAL resolution:
    `metric_all[b]` = Dice(pred > 0, label > 0), `metric_per_cls[b, c-1]` = Dice(pred == c, label == c), c = 1..num_classes,
    0 for an empty prediction (:1463-1472, :1539-1556).  Hausdorff / ASD / Jaccard columns of the reference's [B,4] arrays are
    CPU medpy / SimpleITK work and are not produced.  Returns (metric_all [B], metric_per_cls [B, num_classes], loss, pred)
    -- device tensors, no host sync.  `do_denoise` is the reference's `config.postprocess_mask` (al_trainer.py:1445): the
    morphology of unet_processor.py:72-160 as batched tensor ops on the device (`UnetProcessor.denoise_masks`; parity with cv2 itself is
    unpinned -- cv2 is not importable here).  The model's train / eval mode is restored on every exit path."""
    from transforms.hip import functional_hip as FH
    dev = next(model.parameters()).device
    image = image_batch.to(dev, dtype=torch.float32)
    label = label_batch.to(dev).long()
    was_training = model.training
    model.eval()
    try:
        x = processor.preprocess(image)
        output = model(x)
        pred, _, _ = predict_and_dice(output)
        loss = None
        if loss_fn is not None:
            ll = label
            if pred.shape[-2:] != label.shape[-2:]:
                ll = FH.resize_nearest(label.unsqueeze(1), int(output.shape[-2]), int(output.shape[-1])).squeeze(1)
            loss = loss_fn(output, ll)
        pred = processor.postprocess(pred, label.shape[-2:], do_denoise=do_denoise)
    finally:
        model.train(was_training)
    k1 = num_classes + 1
    dice, counts = label_dice(pred, label, k1)
    n = float(label.shape[-2] * label.shape[-1])
    i0, p0, g0 = counts[:, 0, 0], counts[:, 0, 1], counts[:, 0, 2]
    pf, gf = n - p0, n - g0                       # |pred > 0|, |label > 0|
    inter = n - p0 - g0 + i0                      # |pred > 0 & label > 0| = N - |pred == 0 or label == 0|
    metric_all = torch.where(pf > 0, 2.0 * inter / (pf + gf).clamp_min(1.0), torch.zeros_like(pf))
    return metric_all, dice[:, 1:], loss, pred
