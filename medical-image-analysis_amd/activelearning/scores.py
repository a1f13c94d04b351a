"""Acquisition scores of the active-learning selectors as one fused HIP pass over the logits (softmax + per-image
reduction): entropy (reference `src/activelearning/entropy_selector.py:42-49`), least confidence
(`confidence_selector.py:42-47`), margin (`margin_selector.py:42-48`).  The selectors themselves (pool iteration,
sorting, budget) live in ``activelearning.selectors`` and call ``selector_scores(model(x))``."""
from __future__ import annotations

import ctypes

import torch

from mia_hip import call, lib
from mia_hip.ops import _c_i64, _need_dev, _p, _pix_strides, _stream

ENTROPY, CONFIDENCE, MARGIN = 0, 1, 2


def selector_scores(logits: torch.Tensor, smooth: float = 1e-8) -> torch.Tensor:
    """logits [B,K1,H,W] -> scores [B,3] fp32: (entropy, -max prob, -(top1 - top2)), each averaged like the reference."""
    _need_dev(logits)
    if logits.dtype != torch.float32:
        logits = logits.float()
    st = _pix_strides(logits)
    if st is None:
        logits = logits.contiguous()
        st = _pix_strides(logits)
    b, k1, h, w = logits.shape
    hw = h * w
    slabs = max(1, min(128, hw // 2048))
    ws = torch.empty(lib().mia_selector_scores_workspace(b, slabs), device=logits.device, dtype=torch.float32)
    out = torch.empty((b, 3), device=logits.device, dtype=torch.float32)
    call("mia_selector_scores", _p(logits), b, _c_i64(hw), k1, _c_i64(st[0]), _c_i64(st[1]), _c_i64(st[2]), ctypes.c_float(smooth), slabs, _p(ws), _p(out),
         _stream())
    return out
