"""Batched on-GPU augmentation: the whole per-sample pipeline the reference runs in DataLoader workers
(`src/datasets/fugc/fugc_dataset.py:140-164`: transform at native resolution -> JointResize -> z-score)
as a handful of HIP launches over the minibatch, parameters drawn per sample on the host in the
reference's order.  ``al_train_transforms`` rebuilds the pipeline of `al_trainer.py:670-697`."""
from __future__ import annotations

from typing import Optional

import torch

from .common import ComposeTransform, RandomTransform
from .image_transform import (RandomBrightness, RandomContrast, RandomGamma, RandomGaussianBlur, RandomGaussianNoise,
                              SimulateLowRes)
from .joint_transform import JointResize, MirrorTransform, RandomAffine, RandomElastic, RandomRotation, RandomRotation90
from .normalization import ZScoreNormalize


def al_train_transforms(dataset: str = "fugc", elastic: bool = False) -> ComposeTransform:
    """`ALTrainer._get_train_transform` (al_trainer.py:670-717).  `elastic=True` (build-side option, off by default: the
    reference has no elastic transform) puts `RandomTransform(RandomElastic(), p=0.2)` in front of the reference's stages."""
    if elastic:
        base = al_train_transforms(dataset, elastic=False)
        return ComposeTransform([RandomTransform(RandomElastic(sigma=(0.0, 8.0), grid=(4, 4)), p=0.2)] + base.transforms)
    if dataset.lower() in ("fugc", "busi"):
        return ComposeTransform([
            RandomTransform(RandomAffine(scale=(0.7, 1.4)), p=0.2),
            RandomTransform(RandomAffine(degrees=(-15.0, 15.0)), p=0.2),
            RandomTransform(RandomGaussianNoise(sigma=(0, 0.1)), p=0.1),
            RandomTransform(RandomGaussianBlur(sigma=(0.5, 1.0)), p=0.2),
            RandomTransform(RandomBrightness(brightness=(0.75, 1.25)), p=0.15),
            RandomTransform(RandomContrast(contrast=(0.75, 1.25)), p=0.15),
            RandomTransform(SimulateLowRes(scale=(0.5, 1)), p=0.15),
            RandomTransform(RandomGamma(gamma=(0.7, 1.5)), p=0.1),
        ])
    return ComposeTransform([
        RandomTransform(RandomRotation90(), p=0.5),
        RandomTransform(MirrorTransform((-2, -1)), p=0.5),
        RandomTransform(RandomRotation(degrees=(-20, 20)), p=0.5),
    ])


class BatchedAugment:
    """images [B,C,H0,W0] f32 in [0,1], labels [B,H0,W0] or [B,1,H0,W0] int64 on the GPU ->
    {"image": [B,C,S,S] f32, "label": [B,S,S] int64} ready for ``train_step``."""

    def __init__(self, transform: Optional[ComposeTransform] = None, image_size=None, do_normalize: bool = False,
                 antialias: bool = True):
        self.transform = transform
        self.final = JointResize(image_size, antialias=antialias) if image_size is not None else None
        self.normalize = ZScoreNormalize() if do_normalize else None

    def __call__(self, images: torch.Tensor, labels: torch.Tensor) -> dict:
        b = images.shape[0]
        labels = labels.reshape(b, labels.shape[-2], labels.shape[-1])
        if self.transform is not None:
            shape = tuple(images.shape[1:])
            params = [self.transform.draw(shape) for _ in range(b)]  # sample-major, like sequential per-sample calls
            images, labels = self.transform.apply_batch(images, labels, params)
        if self.final is not None:
            images, labels = self.final.apply_batch(images, labels, [()] * b)
        if self.normalize is not None:
            images, labels = self.normalize.apply_batch(images, labels)
        return {"image": images, "label": labels}
