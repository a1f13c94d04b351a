"""Transform combinators; drop-in for the reference `src/transforms/common.py:12-89`.

Contract kept: ``__call__(data: dict) -> dict`` on ``{"image": [C,H,W] float, "label": [1,H,W] int64}``,
``get_params_dict()``; parameter draws use the global torch CPU generator in the reference's order
(``RandomTransform`` draws ONE uniform before the inner transform draws anything, common.py:27).
Added: every transform splits into ``draw(shape) -> params`` (host) and ``apply_batch(images, labels,
params)`` (HIP kernels over a batch with per-sample parameters) so a whole minibatch is augmented in a
handful of launches (`transforms.gpu_pipeline.BatchedAugment`).  Tensors must be on a HIP device.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import List, Optional

import numpy as np
import torch


class BaseTransform(ABC):
    @abstractmethod
    def get_params_dict(self) -> dict:
        pass

    # -- host: draw this transform's random parameters for ONE sample of shape (C, H, W); None = identity
    def draw(self, shape):
        return ()

    # -- device: images [B,C,H,W] f32, labels [B,H,W] i64 (or None); params = list (len B) of draw() results / None
    def apply_batch(self, images, labels, params):
        raise NotImplementedError

    def __call__(self, data: dict) -> dict:
        image = image_to_tensor(data["image"])
        label = image_to_tensor(data["label"])
        p = self.draw(tuple(image.shape))
        lab_b = label.reshape(1, label.shape[-2], label.shape[-1]) if label is not None else None
        img_o, lab_o = self.apply_batch(image.unsqueeze(0), lab_b, [p])
        data["image"] = img_o[0]
        data["label"] = lab_o.reshape((1,) + tuple(lab_o.shape[-2:])) if lab_o is not None else label
        return data


class RandomTransform(BaseTransform):
    def __init__(self, transform: BaseTransform, p):
        self.p = np.clip(p, 0.0, 1.0)
        self.transform = transform

    def draw(self, shape):
        if torch.rand(1).item() < self.p:
            return self.transform.draw(shape)
        return None

    def apply_batch(self, images, labels, params):
        if all(q is None for q in params):
            return images, labels
        return self.transform.apply_batch(images, labels, params)

    def get_params_dict(self):
        return {RandomTransform.__name__: {"p": self.p, "transform": self.transform.get_params_dict()}}


class RandomChoiceTransform(BaseTransform):
    def __init__(self, transforms: List[BaseTransform], weight: Optional[list] = None):
        self.weight = torch.Tensor(weight) if weight else torch.ones(len(transforms))
        self.transforms = transforms

    def draw(self, shape):
        index = int(torch.multinomial(self.weight, 1).item())
        return (index, self.transforms[index].draw(shape))

    def apply_batch(self, images, labels, params):
        for i, t in enumerate(self.transforms):
            sub = [q[1] if (q is not None and q[0] == i) else None for q in params]
            if any(s is not None for s in sub):
                images, labels = t.apply_batch(images, labels, sub)
        return images, labels

    def get_params_dict(self):
        return {RandomChoiceTransform.__name__: {"weights": self.weight.tolist(),
                                                 "transforms": [t.get_params_dict() for t in self.transforms]}}


class ComposeTransform(BaseTransform):
    def __init__(self, transforms: List[BaseTransform]):
        self.transforms = transforms

    def draw(self, shape):
        out = []
        for t in self.transforms:
            p = t.draw(shape)
            out.append(p)
            shape = t.out_shape(shape, p) if hasattr(t, "out_shape") else shape
        return out

    def apply_batch(self, images, labels, params):
        for i, t in enumerate(self.transforms):
            images, labels = t.apply_batch(images, labels, [None if q is None else q[i] for q in params])
        return images, labels

    def __call__(self, data: dict) -> dict:
        for t in self.transforms:  # sequential, like the reference (common.py:71-74)
            data = t(data)
        return data

    def get_params_dict(self):
        return {ComposeTransform.__name__: {"transforms": [t.get_params_dict() for t in self.transforms]}}


def image_to_tensor(data):
    """Tensors pass through (reference common.py:85-89); PIL / ndarray -> float tensor in [0,1] like F.to_tensor."""
    if data is None or isinstance(data, torch.Tensor):
        return data
    arr = np.asarray(data)
    t = torch.from_numpy(arr)
    if t.ndim == 2:
        t = t[None]
    elif t.ndim == 3:
        t = t.permute(2, 0, 1)
    if t.dtype == torch.uint8:
        t = t.float().div(255)
    return t.contiguous()
