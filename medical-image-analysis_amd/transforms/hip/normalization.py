"""Z-score normalisation on HIP kernels; drop-in for the reference `src/transforms/normalization.py:9-26`
(mean / UNBIASED std over all of C,H,W per sample, std clipped at 1e-8)."""
from __future__ import annotations

import torch

from . import functional_hip as FH
from .common import image_to_tensor


class ZScoreNormalize:
    def __init__(self, target_dtype: torch.dtype = torch.float32):
        self.target_dtype = target_dtype

    def draw(self, shape):
        return ()

    def apply_batch(self, images, labels, params=None):
        ms = FH.sample_stats(images, gray=False)
        return FH.elementwise(images, FH.EW_ZSCORE, mean_std=ms).to(self.target_dtype), labels

    def __call__(self, data: dict) -> dict:
        image = image_to_tensor(data["image"])
        label = image_to_tensor(data["label"])
        data["image"] = self.apply_batch(image.unsqueeze(0), None)[0][0]
        data["label"] = label
        return data
