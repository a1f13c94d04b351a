"""Pre/post-processing around the UNet forward for validation / inference
(reference `src/models/unet/unet_processor.py:11-70`): bilinear resize in, nearest resize back.
``denoise_one_mask`` (reference :72-160, only with ``--postprocess-mask``) is OpenCV morphology on the host in the reference; here
it is the same sequence of operations as batched tensor ops on the device the masks live on (cv2 is not importable in this image:
the arithmetic below follows OpenCV's documented behaviour for uint8 0 / 255 masks and is "parity unpinned" against cv2 itself)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

# cv2.GaussianBlur(ksize, sigma = 0) on uint8 uses the fixed small-kernel table for ksize <= 7, in 8-bit fixed point (x 256):
_SMALL_GAUSS_256 = {1: (256,), 3: (64, 128, 64), 5: (16, 64, 96, 64, 16), 7: (8, 28, 56, 72, 56, 28, 8)}


def _dilate(x: torch.Tensor, r: int) -> torch.Tensor:
    """cv2.dilate with a (2r+1)^2 rectangle: pixels outside the image never contribute (max-pool with -inf padding)."""
    return F.max_pool2d(x, 2 * r + 1, 1, r)


def _erode(x: torch.Tensor, r: int) -> torch.Tensor:
    """cv2.erode with a (2r+1)^2 rectangle: pixels outside the image never contribute."""
    return -F.max_pool2d(-x, 2 * r + 1, 1, r)


def _smooth_threshold(x: torch.Tensor, k: int) -> torch.Tensor:
    """cv2.threshold(cv2.GaussianBlur(m, (k, k), 0), 127, 255, THRESH_BINARY) > 0 for a 0 / 255 mask m = 255 x: the separable blur is
    exact in 8.8 fixed point and rounds half up once at the end, so `blur > 127` <=> sum of the two-dimensional weights over the set
    pixels >= 1/2, evaluated here in integers (weights x 256 per axis, BORDER_REFLECT_101)."""
    if k not in _SMALL_GAUSS_256:
        raise NotImplementedError(f"smooth_kernel={k}: OpenCV's fixed small-kernel table covers 1, 3, 5, 7")
    w = _SMALL_GAUSS_256[k]
    r = k // 2
    xi = x.to(torch.int32)
    if r:
        xi = F.pad(xi.float(), (r, r, r, r), mode="reflect").to(torch.int32)  # 'reflect' excludes the edge pixel = REFLECT_101
    h, wd = x.shape[-2], x.shape[-1]
    rows = sum(w[j] * xi[..., :, j:j + wd] for j in range(k))
    tot = sum(w[i] * rows[..., i:i + h, :] for i in range(k))
    return tot >= 32768


class UnetProcessor:
    def __init__(self, image_size=None, dilate_size: int = 5, erode_size: int = 5, smooth_kernel: int = 7):
        self.dilate_size, self.erode_size, self.smooth_kernel = dilate_size, erode_size, smooth_kernel
        self.mean = None
        self.std = None
        if image_size is not None:
            image_size = [image_size] if isinstance(image_size, int) else list(image_size)
        if image_size and len(image_size) < 2:
            image_size *= 2
        self.image_size = image_size

    def preprocess(self, X: torch.Tensor):
        from transforms.hip import functional_hip as FH
        image = X
        if image.ndim == 3:
            image = image.unsqueeze(0)
        if self.image_size and (self.image_size[0] != image.shape[-2] or self.image_size[1] != image.shape[-1]):
            image = FH.resize_bilinear(image, self.image_size[0], self.image_size[1])
        return image

    def postprocess(self, P: torch.Tensor, ori_shape, do_denoise: bool = False):
        from transforms.hip import functional_hip as FH
        masks = P
        if masks.ndim == 2:
            masks = masks.unsqueeze(0)
        if self.image_size and (ori_shape[0] != masks.shape[-2] or ori_shape[1] != masks.shape[-1]):
            masks = FH.resize_nearest(masks.unsqueeze(1), int(ori_shape[0]), int(ori_shape[1])).squeeze(1)
        else:
            masks = masks.clone()
        if do_denoise:
            masks = self.denoise_masks(masks)
        return masks.to(P.device, dtype=P.dtype)

    def _denoise_binary(self, m: torch.Tensor) -> torch.Tensor:
        """One binary mask [B,1,H,W] of {0,1} floats through pad -> fill_hole (dilate, erode) -> remove_cc (erode, dilate) ->
        unpad -> smoothen_boundary (reference :76-99, :112-160); returns bool."""
        pad = max(self.dilate_size, self.erode_size)
        x = F.pad(m, (pad, pad, pad, pad))  # copyMakeBorder(BORDER_CONSTANT, 0)
        x = _erode(_dilate(x, self.dilate_size), self.erode_size)   # fill_hole
        x = _dilate(_erode(x, self.erode_size), self.dilate_size)   # remove_cc
        x = x[..., pad:x.shape[-2] - pad, pad:x.shape[-1] - pad]
        return _smooth_threshold(x, self.smooth_kernel)

    def denoise_masks(self, masks: torch.Tensor) -> torch.Tensor:
        """`denoise_one_mask` (reference :72-110) for a batch [B,H,W] of label maps: the object mask (all classes) and the mask of
        class 1 are cleaned separately, then the map is rebuilt as 2 everywhere, 1 where the cleaned class-1 mask is set, 0 where
        the cleaned object mask is empty (`num_classes = 2` is hard-coded in the reference)."""
        if masks.ndim == 2:
            return self.denoise_masks(masks.unsqueeze(0))[0]
        obj = self._denoise_binary((masks > 0).float().unsqueeze(1)).squeeze(1)
        cls1 = self._denoise_binary((masks == 1).float().unsqueeze(1)).squeeze(1)
        out = torch.full_like(masks, 2)
        out[cls1] = 1
        out[~obj] = 0
        return out

    def denoise_one_mask(self, P: torch.Tensor) -> torch.Tensor:
        return self.denoise_masks(P.detach())
