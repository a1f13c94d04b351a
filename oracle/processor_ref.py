"""CPU restatement of `UnetProcessor.denoise_one_mask` (TEST INFRASTRUCTURE ONLY; reference
/root/reference/src/models/unet/unet_processor.py:72-160) on numpy + scipy.ndimage instead of cv2.

PARITY UNPINNED: cv2 is not importable in this image and the reference holds no fixture for this function.  The restatement follows
OpenCV's documented behaviour: rectangular dilate / erode ignore pixels outside the image, `GaussianBlur(ksize <= 7, sigma = 0)` on
uint8 uses the fixed small-kernel table in 8.8 fixed point with BORDER_REFLECT_101 and rounds half up, `threshold(127, THRESH_BINARY)`
keeps values > 127."""
import numpy as np
from scipy import ndimage

SMALL_GAUSS = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
               7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}


def dilate(mask: np.ndarray, r: int) -> np.ndarray:  # :147-151
    return ndimage.maximum_filter(mask, size=2 * r + 1, mode="constant", cval=0)


def erode(mask: np.ndarray, r: int) -> np.ndarray:  # :153-157
    return ndimage.minimum_filter(mask, size=2 * r + 1, mode="constant", cval=255)


def smoothen_boundary(mask: np.ndarray, k: int) -> np.ndarray:  # :159-164
    w = np.asarray(SMALL_GAUSS[k], dtype=np.float64)
    rows = ndimage.correlate1d(mask.astype(np.float64), w, axis=1, mode="mirror")  # 8.8 fixed point: exact for these weights
    blur = np.floor(ndimage.correlate1d(rows, w, axis=0, mode="mirror") + 0.5)     # one rounding, half up
    return np.where(blur > 127, 255, 0).astype(np.uint8)


def _clean(binary255: np.ndarray, d: int, e: int, k: int) -> np.ndarray:
    pad = max(d, e)
    m = np.pad(binary255, pad, mode="constant", constant_values=0)  # :122-133
    m = erode(dilate(m, d), e)   # fill_hole :112-116
    m = dilate(erode(m, e), d)   # remove_cc :118-121
    m = m[pad:m.shape[0] - pad, pad:m.shape[1] - pad]
    return smoothen_boundary(m, k)


def denoise_one_mask(mask: np.ndarray, dilate_size: int = 5, erode_size: int = 5, smooth_kernel: int = 7) -> np.ndarray:
    """:72-110 (num_classes = 2 hard-coded there)."""
    obj = _clean(np.where(mask > 0, 255, 0).astype(np.uint8), dilate_size, erode_size, smooth_kernel)
    cls1 = _clean(np.where(mask == 1, 255, 0).astype(np.uint8), dilate_size, erode_size, smooth_kernel)
    out = np.ones_like(mask) * 2
    out[cls1 > 0] = 1
    out[obj == 0] = 0
    return out
