"""Pre/post-processing around the UNet forward for validation / inference
(reference `src/models/unet/unet_processor.py:11-70`): bilinear resize in, nearest resize back.
The cv2 morphology in ``denoise_one_mask`` (reference :72-135, only with ``--postprocess-mask``) is
host-side OpenCV work outside the hot path and is not built."""
from __future__ import annotations

import torch


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
            raise NotImplementedError("postprocess_mask (cv2 morphology, reference unet_processor.py:72-135) is not built")
        return masks.to(P.device, dtype=P.dtype)
