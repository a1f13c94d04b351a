"""CPU restatement of the reference augmentation pipeline (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/src/transforms/common.py:22-74 (combinators),
image_transform.py:15-236 (gamma, contrast, "brightness", noise, blur, low-res),
joint_transform.py:11-206 (resize, rot90, mirror, rotation, crop, affine),
normalization.py:9-26 (z-score).

Two layers:
  * ``apply_*``  -- deterministic arithmetic given already-drawn parameters.
  * ``draw_*``   -- the parameter draws, in the reference's order, from the
    global torch CPU generator (common.py:27, image_transform.py:29,126,159,214,
    joint_transform.py:51,111,140,186).

PINNED against the imported reference (tests/golden/transforms.npz): gamma,
noise, low-res, rot90, mirror, z-score and the combinators' draw order.
PARITY UNPINNED (arithmetic lives in third-party torchvision, un-versioned at
pyproject.toml:19, absent here): affine, rotation, crop, resize, gaussian blur,
contrast/"brightness".  Those follow torchvision's published tensor-path
algorithm (transforms/_functional_tensor.py, functional.py) as described next
to each function, and are pinned by analytic known-answer tests only.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- draws
def draw_apply(p: float) -> bool:
    """RandomTransform: one uniform BEFORE the inner transform draws (common.py:27)."""
    return torch.rand(1).item() < p


def draw_uniform_item(lo: float, hi: float) -> float:
    """``torch.rand(1).item() * (hi - lo) + lo`` (image_transform.py:126-129, :159-162)."""
    return torch.rand(1).item() * (hi - lo) + lo


def draw_gamma(lo: float, hi: float) -> torch.Tensor:
    """image_transform.py:29 -- stays a float32 1-element tensor."""
    return torch.rand(1) * (hi - lo) + lo


def draw_lowres_scales(ndim_spatial: int, lo: float, hi: float) -> List[float]:
    """image_transform.py:213-216."""
    return (torch.rand(ndim_spatial) * (hi - lo) + lo).tolist()


def draw_affine_params(degrees, translate, scale, shear, img_size) -> Tuple[float, Tuple[int, int], float, Tuple[float, float]]:
    """[tv] T.RandomAffine.get_params: angle always drawn first, then translate (if
    given), scale (if given), shear (if given).  The reference passes [h, w] where
    torchvision expects [w, h] (joint_transform.py:186) -- kept as is."""
    angle = float(torch.empty(1).uniform_(float(degrees[0]), float(degrees[1])).item())
    if translate is not None:
        max_dx = float(translate[0] * img_size[0])
        max_dy = float(translate[1] * img_size[1])
        tx = int(round(torch.empty(1).uniform_(-max_dx, max_dx).item()))
        ty = int(round(torch.empty(1).uniform_(-max_dy, max_dy).item()))
        translations = (tx, ty)
    else:
        translations = (0, 0)
    if scale is not None:
        sc = float(torch.empty(1).uniform_(scale[0], scale[1]).item())
    else:
        sc = 1.0
    shear_x = shear_y = 0.0
    if shear is not None:
        shear_x = float(torch.empty(1).uniform_(shear[0], shear[1]).item())
        if len(shear) == 4:
            shear_y = float(torch.empty(1).uniform_(shear[2], shear[3]).item())
    return angle, translations, sc, (shear_x, shear_y)


def draw_rotation(degrees) -> float:
    """[tv] T.RandomRotation.get_params."""
    return float(torch.empty(1).uniform_(float(degrees[0]), float(degrees[1])).item())


def draw_crop(h: int, w: int, th: int, tw: int) -> Tuple[int, int, int, int]:
    """[tv] T.RandomCrop.get_params."""
    if h < th or w < tw:
        raise ValueError(f"Required crop size {(th, tw)} is larger than input image size {(h, w)}")
    if w == tw and h == th:
        return 0, 0, h, w
    i = int(torch.randint(0, h - th + 1, size=(1,)).item())
    j = int(torch.randint(0, w - tw + 1, size=(1,)).item())
    return i, j, th, tw


def draw_contrast(lo: float, hi: float) -> float:
    """[tv] ColorJitter.get_params with only contrast set: randperm(4) first, then
    U(lo, hi) (image_transform.py:56,62)."""
    torch.randperm(4)
    return float(torch.empty(1).uniform_(lo, hi))


def blur_kernel_size(sigma: float, truncate: float = 4.0) -> int:
    """image_transform.py:180-188."""
    c = math.ceil(sigma * truncate + 0.5)
    return c if c % 2 else c - 1


# --------------------------------------------------------------------------- arithmetic
def apply_gamma(image: torch.Tensor, gamma) -> torch.Tensor:
    """image_transform.py:31."""
    g = gamma if isinstance(gamma, torch.Tensor) else torch.tensor([gamma], dtype=torch.float32)
    return torch.pow(image, g)


def apply_noise(image: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """image_transform.py:130-132 with the drawn noise tensor passed explicitly."""
    return torch.clip(image + noise, 0, 1)


def apply_lowres(image: torch.Tensor, scales: Sequence[float]) -> torch.Tensor:
    """image_transform.py:218-225: nearest-exact down, bilinear up."""
    orig = image.shape[1:]
    low = [int(s * i) for s, i in zip(scales, orig)]
    lo = F.interpolate(image[None], low, mode="nearest-exact")
    return F.interpolate(lo, orig, mode="bilinear")[0]


def apply_zscore(image: torch.Tensor) -> torch.Tensor:
    """normalization.py:17-21: mean / unbiased std over all of C,H,W."""
    image = image.to(torch.float32)
    return (image - image.mean()) / image.std().clip(1e-8)


def apply_rot90(x: torch.Tensor, k: int, axes=(-2, -1)) -> torch.Tensor:
    return torch.rot90(x, k, axes)


def apply_mirror(x: torch.Tensor, axes) -> torch.Tensor:
    return torch.flip(x, tuple(axes)) if len(axes) else x


def inverse_affine_matrix(center, angle, translate, scale, shear) -> List[float]:
    """[tv] functional._get_inverse_affine_matrix (inverted=True)."""
    rot = math.radians(angle)
    sx = math.radians(shear[0])
    sy = math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m = [x / scale for x in m]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def _affine_grid_bmm(matrix: Sequence[float], w: int, h: int) -> torch.Tensor:
    """[tv] _functional_tensor._gen_affine_grid with ow=w, oh=h, literally: the grid is an fp32 ``bmm``.  The summation
    order / fusing inside that sgemm belongs to the BLAS kernel the CPU dispatches to, so the last bit of a grid value --
    and with it a nearest-sample tie on about one pixel in a million -- is machine dependent.  Kept to pin `_affine_grid`
    below against torch as run in the dev container (tests/test_oracle_golden.py)."""
    theta = torch.tensor(matrix, dtype=torch.float32).reshape(1, 2, 3)
    d = 0.5
    base = torch.empty(1, h, w, 3, dtype=torch.float32)
    base[..., 0].copy_(torch.linspace(-w * 0.5 + d, w * 0.5 + d - 1, steps=w))
    base[..., 1].copy_(torch.linspace(-h * 0.5 + d, h * 0.5 + d - 1, steps=h).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=torch.float32)
    return base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2)


def _fma32(a, b, c):
    """fp32 fused multiply-add on numpy arrays: the product of two fp32 values is exact in x87 extended precision."""
    import numpy as np
    ld = np.longdouble
    return (a.astype(ld) * b.astype(ld) + c.astype(ld)).astype(np.float32)


def _affine_grid(matrix: Sequence[float], w: int, h: int) -> torch.Tensor:
    """[tv] _gen_affine_grid with the sgemm written out, so the oracle is the same on every machine:
    g = acc(x * t0); acc = fma(y, t1, acc); acc = fma(1, t2, acc) -- k-sequential fused accumulation, which is what MKL's
    sgemm does for this [HW x 3] x [3 x 2] product on the dev container's CPU (bit-equal to `_affine_grid_bmm` there on
    4e7 pixels of random rotations / scales / shears; the unfused order and the reversed order differ on ~1e-6 of them)."""
    import numpy as np
    theta = torch.tensor(matrix, dtype=torch.float32).reshape(2, 3)
    r = (theta / torch.tensor([[0.5 * w], [0.5 * h]], dtype=torch.float32)).numpy()  # one correctly rounded division each
    d = 0.5
    xs = torch.linspace(-w * 0.5 + d, w * 0.5 + d - 1, steps=w).numpy()[None, :]  # step is exactly 1
    ys = torch.linspace(-h * 0.5 + d, h * 0.5 + d - 1, steps=h).numpy()[:, None]
    out = np.empty((1, h, w, 2), dtype=np.float32)
    for j in range(2):
        acc = np.broadcast_to(xs * r[j, 0], (h, w)).astype(np.float32)
        acc = _fma32(np.broadcast_to(ys, (h, w)), np.broadcast_to(r[j, 1], (h, w)), acc)
        out[0, :, :, j] = acc + r[j, 2]
    return torch.from_numpy(out)


def warp_nearest(x: torch.Tensor, matrix: Sequence[float]) -> torch.Tensor:
    """[tv] _apply_grid_transform(mode="nearest", fill=None): float cast for integer
    inputs, grid_sample(nearest, zeros, align_corners=False), round + cast back."""
    c, h, w = x.shape
    grid = _affine_grid(matrix, w, h)
    xf = x if x.is_floating_point() else x.to(torch.float32)
    out = F.grid_sample(xf[None], grid, mode="nearest", padding_mode="zeros", align_corners=False)[0]
    if not x.is_floating_point():
        out = torch.round(out).to(x.dtype)
    return out


def apply_affine(x: torch.Tensor, angle: float, translate, scale: float, shear) -> torch.Tensor:
    """[tv] F.affine tensor path: centre [0,0] in the centred base-grid frame,
    interpolation NEAREST, fill 0 (joint_transform.py:189-190)."""
    m = inverse_affine_matrix([0.0, 0.0], angle, [1.0 * t for t in translate], scale, shear)
    return warp_nearest(x, m)


def apply_rotate(x: torch.Tensor, angle: float) -> torch.Tensor:
    """[tv] F.rotate tensor path (expand=False): inverse matrix of -angle
    (joint_transform.py:113-114)."""
    m = inverse_affine_matrix([0.0, 0.0], -angle, [0.0, 0.0], 1.0, [0.0, 0.0])
    return warp_nearest(x, m)


def apply_elastic(x: torch.Tensor, disp: torch.Tensor) -> torch.Tensor:
    """Elastic deformation -- NO reference counterpart: restatement of the build's own spec
    (medical-image-analysis_amd/transforms/hip/joint_transform.py::RandomElastic, csrc/augment.hip::elastic_warp_kernel).
    x: [C, H, W] float image (bilinear, zero outside) or integer label map (nearest, zero outside); disp: [2, gh, gw] fp32
    control-point displacements in pixels (0 = x, 1 = y).  Every step is a separate fp32 torch op, in the kernel's order."""
    c, h, w = x.shape
    gh, gw = disp.shape[1], disp.shape[2]
    f32 = torch.float32
    su = torch.tensor((gw - 1), dtype=f32) / torch.tensor((w - 1), dtype=f32) if w > 1 else torch.tensor(0.0)
    sv = torch.tensor((gh - 1), dtype=f32) / torch.tensor((h - 1), dtype=f32) if h > 1 else torch.tensor(0.0)
    xs = torch.arange(w, dtype=f32)
    ys = torch.arange(h, dtype=f32)
    u, v = xs * su, ys * sv
    j0 = u.to(torch.int64).clamp(0, gw - 2)
    i0 = v.to(torch.int64).clamp(0, gh - 2)
    tu = (u - j0.to(f32))[None, :]
    tv = (v - i0.to(f32))[:, None]
    d = []
    for k in range(2):
        g = disp[k].to(f32)
        g00, g01 = g[i0][:, j0], g[i0][:, j0 + 1]
        g10, g11 = g[i0 + 1][:, j0], g[i0 + 1][:, j0 + 1]
        top = (1.0 - tu) * g00 + tu * g01
        bot = (1.0 - tu) * g10 + tu * g11
        d.append((1.0 - tv) * top + tv * bot)
    sx = xs[None, :] + d[0]
    sy = ys[:, None] + d[1]
    if not x.is_floating_point():
        rx, ry = torch.round(sx), torch.round(sy)  # half to even, like v_rndne_f32
        inside = (rx >= 0) & (rx <= w - 1) & (ry >= 0) & (ry <= h - 1)
        xi, yi = rx.clamp(0, w - 1).long(), ry.clamp(0, h - 1).long()
        out = x[:, yi, xi]
        return torch.where(inside[None], out, torch.zeros_like(out))
    fx, fy = torch.floor(sx), torch.floor(sy)
    ax, ay = sx - fx, sy - fy
    xi, yi = fx.long(), fy.long()

    def at(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        val = x[:, yy.clamp(0, h - 1), xx.clamp(0, w - 1)]
        return torch.where(ok[None], val, torch.zeros_like(val))

    top = (1.0 - ax) * at(yi, xi) + ax * at(yi, xi + 1)
    bot = (1.0 - ax) * at(yi + 1, xi) + ax * at(yi + 1, xi + 1)
    return (1.0 - ay) * top + ay * bot


def apply_crop(x: torch.Tensor, i: int, j: int, h: int, w: int) -> torch.Tensor:
    return x[..., i:i + h, j:j + w]


def gaussian_kernel1d(ksize: int, sigma: float) -> torch.Tensor:
    """[tv] _get_gaussian_kernel1d."""
    half = (ksize - 1) * 0.5
    x = torch.linspace(-half, half, steps=ksize)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    return pdf / pdf.sum()


def apply_gaussian_blur(image: torch.Tensor, ksize: int, sigma: float) -> torch.Tensor:
    """[tv] F.gaussian_blur tensor path: outer-product 2-D kernel, reflect pad k//2,
    depthwise conv (image_transform.py:166-170)."""
    k1 = gaussian_kernel1d(ksize, sigma)
    k2 = torch.mm(k1[:, None], k1[None, :])
    c = image.shape[0]
    kern = k2.expand(c, 1, ksize, ksize)
    pad = ksize // 2
    x = F.pad(image[None], [pad, pad, pad, pad], mode="reflect")
    return F.conv2d(x, kern, groups=c)[0]


def apply_contrast(image: torch.Tensor, factor: float) -> torch.Tensor:
    """[tv] adjust_contrast tensor path: blend with the mean of the (grayscale)
    image, clamp to [0,1] (image_transform.py:62 and :93 -- "brightness" too)."""
    c = image.shape[0]
    if c == 3:
        r, g, b = image.unbind(0)
        gray = (0.2989 * r + 0.587 * g + 0.114 * b).to(image.dtype).unsqueeze(0)
        mean = gray.mean(dim=(-3, -2, -1), keepdim=True)
    else:
        mean = image.mean(dim=(-3, -2, -1), keepdim=True)
    return (factor * image + (1.0 - factor) * mean).clamp(0, 1.0)


def apply_resize_image(image: torch.Tensor, size: Sequence[int], antialias: bool = False) -> torch.Tensor:
    """[tv] F.resize(BILINEAR) tensor path = interpolate(bilinear, align_corners=False,
    antialias=<torchvision-version dependent>) (joint_transform.py:24)."""
    return F.interpolate(image[None], size=list(size), mode="bilinear", align_corners=False,
                         antialias=antialias)[0]


def apply_resize_label(label: torch.Tensor, size: Sequence[int]) -> torch.Tensor:
    """[tv] F.resize(NEAREST) tensor path: integer inputs go through float32
    (joint_transform.py:25)."""
    lf = label.to(torch.float32) if not label.is_floating_point() else label
    out = F.interpolate(lf[None], size=list(size), mode="nearest")[0]
    if not label.is_floating_point():
        out = torch.round(out).to(label.dtype)
    return out


# --------------------------------------------------------------------------- al_train pipeline
def al_train_fugc_pipeline(image: torch.Tensor, label: torch.Tensor, record: Optional[list] = None):
    """The 8-stage ComposeTransform ``al_train`` builds for fugc/busi
    (al_trainer.py:674-697), drawing from the global torch generator in the
    reference's order.  ``record`` collects (name, params) for replay on the GPU."""
    rec = record if record is not None else []
    _, h, w = image.shape
    if draw_apply(0.2):  # RandomAffine(scale=(0.7,1.4))
        a, t, s, sh = draw_affine_params([0.0, 0.0], None, [0.7, 1.4], None, [h, w])
        image, label = apply_affine(image, a, t, s, sh), apply_affine(label, a, t, s, sh)
        rec.append(("affine", (a, t, s, sh)))
    if draw_apply(0.2):  # RandomAffine(degrees=(-15,15))
        a, t, s, sh = draw_affine_params([-15, 15], None, None, None, [h, w])
        image, label = apply_affine(image, a, t, s, sh), apply_affine(label, a, t, s, sh)
        rec.append(("affine", (a, t, s, sh)))
    if draw_apply(0.1):  # RandomGaussianNoise(sigma=(0,0.1))
        sigma = draw_uniform_item(0.0, 0.1)
        noise = torch.normal(0, sigma, size=image.shape)
        image = apply_noise(image, noise)
        rec.append(("noise", (sigma, noise)))
    if draw_apply(0.2):  # RandomGaussianBlur(sigma=(0.5,1.0))
        sigma = draw_uniform_item(0.5, 1.0)
        k = blur_kernel_size(sigma)
        image = apply_gaussian_blur(image, k, sigma)
        rec.append(("blur", (k, sigma)))
    if draw_apply(0.15):  # RandomBrightness == second contrast jitter
        f = draw_contrast(0.75, 1.25)
        image = apply_contrast(image, f)
        rec.append(("contrast", (f,)))
    if draw_apply(0.15):  # RandomContrast
        f = draw_contrast(0.75, 1.25)
        image = apply_contrast(image, f)
        rec.append(("contrast", (f,)))
    if draw_apply(0.15):  # SimulateLowRes(scale=(0.5,1))
        sc = draw_lowres_scales(2, 0.5, 1.0)
        image = apply_lowres(image, sc)
        rec.append(("lowres", (sc,)))
    if draw_apply(0.1):  # RandomGamma(gamma=(0.7,1.5))
        g = draw_gamma(0.7, 1.5)
        image = apply_gamma(image, g)
        rec.append(("gamma", (float(g),)))
    return image, label, rec
