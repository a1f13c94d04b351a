"""Geometric / joint transforms on HIP kernels; drop-in for the reference `src/transforms/joint_transform.py`.
Image and label are moved by the SAME sampled parameters in one launch.  torchvision's host-side
parameter math (T.RandomAffine.get_params, F._get_inverse_affine_matrix, T.RandomCrop.get_params) is
restated here because torchvision is a third-party dependency of the reference, not part of it."""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import torch

from . import functional_hip as FH
from .common import BaseTransform

IDENTITY = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0]


def inverse_affine_matrix(center, angle, translate, scale, shear) -> List[float]:
    """torchvision functional._get_inverse_affine_matrix."""
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d / scale, -b / scale, 0.0, -c / scale, a / scale, 0.0]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


class JointResize(BaseTransform):
    """F.resize(image, BILINEAR) / F.resize(label, NEAREST) (reference :24-25).  ``antialias`` mirrors torchvision's
    version-dependent tensor-path default (True since 0.17; the reference pins no version)."""

    def __init__(self, image_size, antialias: bool = True):
        if isinstance(image_size, int):
            image_size = (image_size, image_size)
        if len(image_size) < 2:
            image_size = image_size * 2
        self.image_size = list(image_size)
        self.antialias = antialias

    def out_shape(self, shape, p):
        return (shape[0], self.image_size[0], self.image_size[1])

    def apply_batch(self, images, labels, params):
        oh, ow = self.image_size
        img = FH.resize_bilinear(images, oh, ow, antialias=self.antialias)
        lab = FH.resize_nearest(labels, oh, ow) if labels is not None else None
        return img, lab

    def get_params_dict(self):
        return {JointResize.__name__: {"image_size": self.image_size}}


class RandomRotation90(BaseTransform):
    def __init__(self, axes: Tuple[int, int] = (-2, -1)):
        assert axes[0] != axes[1]
        if tuple(axes) != (-2, -1):
            raise NotImplementedError("RandomRotation90 on the MI355X path rotates in the (H, W) plane")
        self.axes = axes

    def draw(self, shape):
        return (int(torch.randint(0, 4, (1,)).item()),)

    def out_shape(self, shape, p):
        return (shape[0], shape[2], shape[1]) if p and p[0] % 2 else shape

    def apply_batch(self, images, labels, params):
        ks = {p[0] for p in params if p is not None}
        if len(ks) > 1 or any(p is None for p in params) and ks - {0}:
            # per-sample quarter turns change shapes independently: handle sample by sample
            outs = [self.apply_batch(images[i:i + 1], None if labels is None else labels[i:i + 1], [params[i]])
                    if params[i] is not None else (images[i:i + 1], None if labels is None else labels[i:i + 1])
                    for i in range(len(params))]
            return torch.cat([o[0] for o in outs]), (None if labels is None else torch.cat([o[1] for o in outs]))
        k = ks.pop() if ks else 0
        return FH.rot90_flip(images, k), (FH.rot90_flip(labels, k) if labels is not None else None)

    def get_params_dict(self):
        return {RandomRotation90.__name__: {"axes": self.axes}}


class MirrorTransform(BaseTransform):
    def __init__(self, axes):
        if not isinstance(axes, Sequence):
            axes = tuple([axes])
        self.axes = axes

    def apply_batch(self, images, labels, params):
        if len(self.axes) == 0:
            return images, labels
        ax = {a % 3 for a in self.axes}  # per-sample tensors are [C, H, W]
        if 0 in ax:
            raise NotImplementedError("MirrorTransform over the channel axis is not built")
        fh, fw = 1 in ax, 2 in ax
        return FH.rot90_flip(images, 0, fh, fw), (FH.rot90_flip(labels, 0, fh, fw) if labels is not None else None)

    def get_params_dict(self):
        return {MirrorTransform.__name__: {"allowed_axes": self.axes}}


class RandomRotation(BaseTransform):
    def __init__(self, degrees):
        if not isinstance(degrees, Sequence):
            degrees = [-degrees, degrees]
        self.degrees = list(degrees)

    def draw(self, shape):
        angle = float(torch.empty(1).uniform_(float(self.degrees[0]), float(self.degrees[1])).item())
        return (inverse_affine_matrix([0.0, 0.0], -angle, [0.0, 0.0], 1.0, [0.0, 0.0]),)  # F.rotate

    def apply_batch(self, images, labels, params):
        return FH.affine_nearest(images, labels, [p[0] if p else IDENTITY for p in params], [p is not None for p in params])

    def get_params_dict(self):
        return {RandomRotation.__name__: {"degrees": self.degrees}}


class RandomCrop2D(BaseTransform):
    def __init__(self, crop):
        if not isinstance(crop, (List, Tuple)):
            crop = (crop, crop)
        self.crop = crop

    def draw(self, shape):
        _, h, w = shape
        th, tw = self.crop
        if h < th or w < tw:
            raise ValueError(f"Required crop size {(th, tw)} is larger than input image size {(h, w)}")
        if w == tw and h == th:
            return (0, 0, h, w)
        i = int(torch.randint(0, h - th + 1, size=(1,)).item())
        j = int(torch.randint(0, w - tw + 1, size=(1,)).item())
        return (i, j, th, tw)

    def out_shape(self, shape, p):
        return (shape[0], self.crop[0], self.crop[1])

    def apply_batch(self, images, labels, params):
        th, tw = self.crop
        top, left = [], []
        for p in params:  # None (skipped by an enclosing RandomTransform) cannot keep the old size in a batch: crop at 0,0
            i, j, h, w = p if p is not None else (0, 0, th, tw)
            if (h, w) != (th, tw):
                raise ValueError(f"RandomCrop2D: window {(h, w)} differs from the configured crop {(th, tw)}")
            top.append(i)
            left.append(j)
        return FH.crop(images, top, left, th, tw), (FH.crop(labels, top, left, th, tw) if labels is not None else None)

    def get_params_dict(self):
        return {RandomCrop2D.__name__: {"crop": self.crop}}


class RandomElastic(BaseTransform):
    """Elastic deformation of image and label -- a BUILD-SIDE ADDITION: `BASELINE.json.north_star` names an elastic
    augmentation kernel, the reference has none (`grep -ri elastic src` -> 0 hits), so there is no reference
    behaviour to match ("parity unpinned" by construction) and it is part of no `al_train` pipeline unless asked for
    (`al_train_transforms(..., elastic=True)`).  Spec (U-Net paper's scheme; kernel `mia_elastic_warp`, CPU restatement
    `oracle/transforms_ref.py::apply_elastic`):

    * draw: `sigma = U(sigma[0], sigma[1])` by `torch.rand(1)`, then `D = torch.randn(2, grid[0], grid[1]) * sigma` -- displacement
      vectors in pixels (component 0 = x, 1 = y) on a coarse grid of control points spanning the image corner to corner;
    * per-pixel displacement = bilinear interpolation of D (control point (i, j) sits at pixel (i*(H-1)/(gh-1), j*(W-1)/(gw-1)));
    * image: bilinear sample at (x + dx, y + dy), zero outside; label: nearest source pixel (round half even), zero outside.
    """

    def __init__(self, sigma=(0.0, 8.0), grid=(4, 4)):
        if not isinstance(sigma, Sequence):
            sigma = (0.0, float(sigma))
        if not isinstance(grid, Sequence):
            grid = (int(grid), int(grid))
        if grid[0] < 2 or grid[1] < 2:
            raise ValueError("RandomElastic: the control grid needs at least 2 x 2 points")
        self.sigma, self.grid = [float(sigma[0]), float(sigma[1])], [int(grid[0]), int(grid[1])]

    def draw(self, shape):
        s = float(torch.rand(1).item() * (self.sigma[1] - self.sigma[0]) + self.sigma[0])
        return (torch.randn(2, self.grid[0], self.grid[1]) * s, s)

    def apply_batch(self, images, labels, params):
        zero = torch.zeros(2, self.grid[0], self.grid[1])
        disp = torch.stack([p[0] if p is not None else zero for p in params]).numpy()  # host array: rides in the batch's one upload
        return FH.elastic_warp(images, labels, disp, [p is not None for p in params])

    def get_params_dict(self):
        return {RandomElastic.__name__: {"sigma": self.sigma, "grid": self.grid}}


class RandomAffine(BaseTransform):
    def __init__(self, degrees=0.0, translate=None, scale=None, shear=None):
        if not isinstance(degrees, Sequence):
            degrees = [-degrees, degrees]
        self.degrees = list(degrees)
        self.translate = list(translate) if translate else None
        self.scale = list(scale) if scale else None
        if shear:
            if not isinstance(shear, Sequence):
                shear = [-shear, shear]
            self.shear = list(shear)
        else:
            self.shear = None

    def draw(self, shape):
        """T.RandomAffine.get_params draw order: angle (always), translate, scale, shear -- each only if configured.
        The reference passes [h, w] as img_size (joint_transform.py:186); kept."""
        _, h, w = shape
        img_size = [h, w]
        angle = float(torch.empty(1).uniform_(float(self.degrees[0]), float(self.degrees[1])).item())
        if self.translate is not None:
            max_dx, max_dy = float(self.translate[0] * img_size[0]), float(self.translate[1] * img_size[1])
            tx = int(round(torch.empty(1).uniform_(-max_dx, max_dx).item()))
            ty = int(round(torch.empty(1).uniform_(-max_dy, max_dy).item()))
            translations = (tx, ty)
        else:
            translations = (0, 0)
        sc = float(torch.empty(1).uniform_(self.scale[0], self.scale[1]).item()) if self.scale is not None else 1.0
        shear_x = shear_y = 0.0
        if self.shear is not None:
            shear_x = float(torch.empty(1).uniform_(self.shear[0], self.shear[1]).item())
            if len(self.shear) == 4:
                shear_y = float(torch.empty(1).uniform_(self.shear[2], self.shear[3]).item())
        m = inverse_affine_matrix([0.0, 0.0], angle, [1.0 * t for t in translations], sc, [shear_x, shear_y])
        return (m, angle, translations, sc, (shear_x, shear_y))

    def apply_batch(self, images, labels, params):
        return FH.affine_nearest(images, labels, [p[0] if p else IDENTITY for p in params], [p is not None for p in params])

    def get_params_dict(self):
        return {RandomAffine.__name__: {"degrees": self.degrees, "translate": self.translate, "scale": self.scale,
                                        "shear": self.shear}}
