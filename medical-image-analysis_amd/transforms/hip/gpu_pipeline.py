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
                 antialias: bool = True, inplace: bool = False):
        """inplace=True: `images` / `labels` may be overwritten (a batch fresh from the loader); False (default): the stages
        work on a copy (one device-to-device copy of the batch).  Either way every stage touches only the samples it was drawn
        for (functional_hip.set_selective): al_train's stages fire with p = 0.1 .. 0.2 per sample (al_trainer.py:674-697)."""
        self.inplace = inplace
        self.transform = transform
        self.final = JointResize(image_size, antialias=antialias) if image_size is not None else None
        self.normalize = ZScoreNormalize() if do_normalize else None
        self._arena = None

    def __call__(self, images: torch.Tensor, labels: torch.Tensor) -> dict:
        """Parameters are drawn on the host in the reference's order, then the stages run twice: a dry pass on `meta` tensors
        that only collects every stage's per-sample parameter arrays, ONE pinned asynchronous upload of all of them, and the
        real pass (functional_hip.ParamArena) -- no per-stage copies, no host/device synchronisation."""
        from . import functional_hip as FH
        b = images.shape[0]
        labels = labels.reshape(b, labels.shape[-2], labels.shape[-1])
        params = None
        if self.transform is not None:
            shape = tuple(images.shape[1:])
            params = [self.transform.draw(shape) for _ in range(b)]  # sample-major, like sequential per-sample calls
        if self._arena is None or self._arena.device != images.device:
            self._arena = FH.ParamArena(images.device)
        arena = self._arena
        FH.set_arena(arena)
        fires = params is not None and any(q is not None and any(x is not None for x in q) for q in params)
        if fires:  # the stages modify their batch in place (selected samples only): work on a copy unless told otherwise
            img_f, lab_l = images.float(), labels.long()  # no-ops for the usual fp32 / int64 inputs
            if not self.inplace:
                img_f = img_f.clone() if img_f.data_ptr() == images.data_ptr() else img_f  # (hipMemcpy D2D)
                lab_l = lab_l.clone() if lab_l.data_ptr() == labels.data_ptr() else lab_l
            images, labels = img_f.contiguous(), lab_l.contiguous()
        FH.set_selective(bool(fires))
        try:
            arena.begin_dry()
            rng_before = torch.get_rng_state()
            self._run(torch.empty(images.shape, dtype=images.dtype, device="meta"),
                      torch.empty(labels.shape, dtype=labels.dtype, device="meta"), params, b)
            # contract: `apply_batch` is pure w.r.t. host state -- every random draw happens in `draw()` above, in the
            # reference's order; a stage that drew inside apply_batch would draw twice (dry + real) and shift the stream
            if not torch.equal(rng_before, torch.get_rng_state()):
                raise RuntimeError("BatchedAugment: a stage's apply_batch consumed the torch CPU generator; draws belong in draw()")
            arena.upload()
            images, labels, nbytes = self._run(images, labels, params, b)
        finally:
            arena.dry = False
            FH.set_arena(None)
            FH.set_selective(False)
        return {"image": images, "label": labels, "_bytes": nbytes}

    def _run(self, images, labels, params, b):
        nbytes = 0
        if self.transform is not None:
            nbytes += self._stage_bytes(params, images, labels)
            images, labels = self.transform.apply_batch(images, labels, params)
        if self.final is not None:
            h0, w0 = images.shape[-2:]
            images, labels = self.final.apply_batch(images, labels, [()] * b)
            nbytes += b * (4 * images.shape[1] + 8) * (h0 * w0 + images.shape[-2] * images.shape[-1])
        if self.normalize is not None:
            nbytes += 3 * images.numel() * 4
            images, labels = self.normalize.apply_batch(images, labels)
        return images, labels, nbytes

    def _stage_bytes(self, params, images, labels) -> int:
        """Algorithmic bytes of the stages that run on this batch (a stage reads its image (+ label) once and writes
        it once), counted per SELECTED sample: a stage drawn for k of the B samples streams k samples."""
        b, c, h, w = images.shape
        img_b, lab_b = c * h * w * 4, h * w * 8  # per sample
        total = 0
        stages = getattr(self.transform, "transforms", [])
        for i, t in enumerate(stages):
            k = sum(1 for q in params if q is not None and q[i] is not None)
            if k == 0:
                continue
            inner = getattr(t, "transform", t)
            geometric = isinstance(inner, (RandomAffine, RandomElastic, RandomRotation, RandomRotation90, MirrorTransform))
            total += k * (2 * img_b + (2 * lab_b if geometric else 0))
            if isinstance(inner, (RandomContrast, RandomBrightness)):
                total += k * img_b  # the mean pass
        return total
