"""Intensity transforms on HIP kernels; drop-in for the reference `src/transforms/image_transform.py`.
Constructor signatures, parameter draws (global torch CPU RNG, same order) and ``get_params_dict`` are
the reference's; labels pass through untouched."""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import torch

from . import functional_hip as FH
from .common import BaseTransform


def _on(params):
    return [p is not None for p in params]


class RandomGamma(BaseTransform):
    def __init__(self, gamma):
        if not isinstance(gamma, Sequence):
            gamma = [gamma, gamma]
        self.gamma = list(gamma)

    def draw(self, shape):
        g = torch.rand(1) * (self.gamma[1] - self.gamma[0]) + self.gamma[0]  # image_transform.py:29
        return (float(g),)

    def apply_batch(self, images, labels, params):
        return FH.elementwise(images, FH.EW_GAMMA, p0=[p[0] if p else 1.0 for p in params], apply=_on(params)), labels

    def get_params_dict(self):
        return {RandomGamma.__name__: {"gamma": self.gamma}}


class _ContrastJitter(BaseTransform):
    """torchvision ColorJitter(contrast=(lo, hi)) tensor path: randperm(4), U(lo,hi), blend with the image mean."""

    def _range(self, v):
        if not isinstance(v, Sequence):
            v = (max(1.0 - v, 0.0), 1.0 + v)
        return v

    def draw(self, shape):
        lo, hi = self._lohi
        torch.randperm(4)  # ColorJitter.get_params draws the op order first
        if float(lo) == float(hi) == 1.0:
            return (1.0,)
        return (float(torch.empty(1).uniform_(float(lo), float(hi))),)

    def apply_batch(self, images, labels, params):
        ms = FH.sample_stats(images, gray=(images.shape[1] == 3), apply=_on(params))
        return FH.elementwise(images, FH.EW_CONTRAST, p0=[p[0] if p else 1.0 for p in params], mean_std=ms, apply=_on(params)), labels


class RandomContrast(_ContrastJitter):
    def __init__(self, contrast):
        self.contrast = self._range(contrast)
        self._lohi = self.contrast

    def get_params_dict(self):
        return {RandomContrast.__name__: {"contrast": self.contrast}}


class RandomBrightness(_ContrastJitter):
    """The reference builds ``T.ColorJitter(contrast=self.brightness)`` (image_transform.py:87): this IS a second
    contrast jitter, not a brightness change.  Kept as is for parity."""

    def __init__(self, brightness):
        self.brightness = self._range(brightness)
        self._lohi = self.brightness

    def get_params_dict(self):
        return {RandomBrightness.__name__: {"brightness": self.brightness}}


class RandomGaussianNoise(BaseTransform):
    """``exact_rng=True`` draws the noise tensor from the torch CPU generator exactly like the reference
    (image_transform.py:130) and uploads it; the default generates it on the device (Philox4x32-10)."""

    exact_rng = False

    def __init__(self, sigma):
        if not isinstance(sigma, Sequence):
            sigma = [sigma, sigma]
        self.sigma = list(sigma)
        self._ctr = 0

    def draw(self, shape):
        sigma = torch.rand(1).item() * (self.sigma[1] - self.sigma[0]) + self.sigma[0]
        if self.exact_rng:
            return (sigma, torch.normal(0, sigma, size=shape))
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        return (sigma, seed)

    def apply_batch(self, images, labels, params):
        on = _on(params)
        if any(p is not None and isinstance(p[1], torch.Tensor) for p in params):
            noise = torch.stack([p[1] if p is not None else torch.zeros(images.shape[1:]) for p in params]).to(images.device)
            return FH.elementwise(images, FH.EW_NOISE, aux=noise, apply=on), labels
        seed = next((p[1] for p in params if p is not None), 0)
        if not FH._dry():  # the dry (parameter-collecting) pass of BatchedAugment must leave host state alone
            self._ctr += 1
        return FH.noise_clip(images, [p[0] if p else 0.0 for p in params], seed, self._ctr, on), labels

    def get_params_dict(self):
        return {RandomGaussianNoise.__name__: {"sigma": self.sigma}}


class RandomGaussianBlur(BaseTransform):
    def __init__(self, sigma):
        if not isinstance(sigma, Sequence):
            sigma = [sigma, sigma]
        self.sigma = list(sigma)

    def draw(self, shape):
        sigma = torch.rand(1).item() * (self.sigma[1] - self.sigma[0]) + self.sigma[0]
        return (sigma, self._get_kernel_size(sigma))

    def apply_batch(self, images, labels, params):
        return FH.gaussian_blur(images, [p[0] if p else 1.0 for p in params], [p[1] if p else 1 for p in params], _on(params)), labels

    def _get_kernel_size(self, sigma: float, truncate: float = 4.0):
        return self._round_to_odd(sigma * truncate + 0.5)

    def _round_to_odd(self, x: float):
        c = math.ceil(x)
        return c if c % 2 else c - 1

    def get_params_dict(self):
        return {RandomGaussianBlur.__name__: {"sigma": self.sigma}}


class SimulateLowRes(BaseTransform):
    def __init__(self, scale):
        if not isinstance(scale, Sequence):
            scale = [scale, scale]
        self.scale = list(scale)
        self.upmodes = {1: "linear", 2: "bilinear", 3: "trilinear"}

    def draw(self, shape):
        if len(shape) != 3:
            raise NotImplementedError("SimulateLowRes on the MI355X path handles [C, H, W] images")
        scales = (torch.rand(2) * (self.scale[1] - self.scale[0]) + self.scale[0]).tolist()
        return tuple(int(s * i) for s, i in zip(scales, shape[1:]))

    def apply_batch(self, images, labels, params):
        h, w = images.shape[-2:]
        return FH.lowres(images, [p if p else (h, w) for p in params], _on(params)), labels

    def get_params_dict(self):
        return {SimulateLowRes.__name__: {"scale": self.scale}}
