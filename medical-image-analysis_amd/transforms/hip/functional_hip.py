"""Tensor-level wrappers of the augmentation / resize / normalisation kernels (libmia_hip).

Batched: images ``[B, C, H, W]`` fp32, labels ``[B, H, W]`` int64 on a HIP device; per-sample
parameters as Python lists (uploaded as tiny device arrays); ``apply`` = list of bools (None = all).
"""
from __future__ import annotations

import ctypes
import threading
import time
from typing import Optional, Sequence

import torch

from mia_hip import MiaError, call, lib
from mia_hip.ops import _c_float, _c_i64, _need_dev, _p, _stream

import numpy as np

EW_GAMMA, EW_CONTRAST, EW_NOISE, EW_ZSCORE = 0, 1, 2, 3


class ParamArena:
    """ONE host -> device parameter upload per augmented batch.

    The per-sample parameters of a pipeline (affine matrices, sigmas, kernel sizes, apply flags, displacement grids ...) are
    produced stage by stage inside ``apply_batch``; uploaded one by one (`torch.tensor(list, device=...)`) each is a pageable,
    SYNCHRONOUS copy that also makes the host wait for everything queued on the stream -- 17 copies per batch and a host
    that can never run ahead of the GPU.  With an arena the pipeline runs twice: a DRY pass on `meta` tensors (no memory,
    no launch) in which every parameter array is only registered, then all arrays travel in one pinned, asynchronous copy,
    and the REAL pass hands each request its slice of the device buffer in the same order."""

    RING = 4  # pinned staging buffers in flight (the host may run this many batches ahead of the device)

    def __init__(self, device):
        self.device = device
        self.dry = False
        self.items = []
        self.views = []
        self.k = 0
        self._slots = [None] * self.RING  # (pinned host uint8, device uint8, event)
        self._turn = 0
        self.wait_s = 0.0  # seconds the host spent blocked on a staging buffer the device had not consumed yet (GPU-bound loop)

    def begin_dry(self):
        self.dry, self.items, self.views, self.k = True, [], [], 0

    def upload(self):
        """End of the dry pass: pack the registered arrays (16-byte aligned) and issue the one copy."""
        self.dry = False
        offs, total = [], 0
        for a in self.items:
            offs.append(total)
            total += (a.nbytes + 15) // 16 * 16
        if total == 0:
            return
        slot = self._slots[self._turn]
        if slot is None or slot[0].numel() < total:
            cap = max(4096, 2 * total)
            slot = (torch.empty(cap, dtype=torch.uint8).pin_memory(), torch.empty(cap, dtype=torch.uint8, device=self.device),
                    torch.cuda.Event())
            self._slots[self._turn] = slot
        else:
            t0 = time.perf_counter()
            slot[2].synchronize()  # the copy that last read this pinned buffer has completed (normally long ago)
            self.wait_s += time.perf_counter() - t0
        host, dev, ev = slot
        hv = host.numpy()
        for a, o in zip(self.items, offs):
            hv[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
        dev[:total].copy_(host[:total], non_blocking=True)
        ev.record()
        self._turn = (self._turn + 1) % self.RING
        td = {np.dtype(np.float32): torch.float32, np.dtype(np.int32): torch.int32}
        self.views = [dev[o:o + a.nbytes].view(td[a.dtype]).reshape(a.shape) for a, o in zip(self.items, offs)]
        self.k = 0

    def take(self, arr: np.ndarray) -> torch.Tensor:
        if self.dry:
            self.items.append(np.ascontiguousarray(arr))
            return torch.empty(arr.shape, dtype=torch.float32 if arr.dtype == np.float32 else torch.int32, device="meta")
        if self.k >= len(self.views):
            raise MiaError("ParamArena: the real pass asked for more parameter arrays than the dry pass registered")
        v, reg = self.views[self.k], self.items[self.k]
        # the k-th request of the real pass must be the k-th array of the dry pass: same shape, dtype AND bytes (a stage whose
        # apply_batch is not pure w.r.t. host state -- RNG draws, counters -- would otherwise run on the wrong parameters)
        if tuple(reg.shape) != tuple(arr.shape) or reg.dtype != arr.dtype or not np.array_equal(reg, arr):
            raise MiaError("ParamArena: the real pass asked for a different parameter array than the dry pass "
                           "(apply_batch must be pure with respect to host state)")
        self.k += 1
        return v


_TLS = threading.local()  # the active arena is per THREAD: two pipelines in two worker threads never share one


def set_arena(arena: Optional[ParamArena]) -> None:
    """Set by transforms.gpu_pipeline.BatchedAugment around one batch (calling thread only)."""
    _TLS.arena = arena


def _arena() -> Optional[ParamArena]:
    return getattr(_TLS, "arena", None)


def _dry() -> bool:
    a = _arena()
    return a is not None and a.dry


def dev_array(arr: np.ndarray, dev) -> torch.Tensor:
    """float32 / int32 host array -> device tensor: a slice of the batch's single upload when an arena is active."""
    a = _arena()
    if a is not None:
        return a.take(arr)
    return torch.from_numpy(np.ascontiguousarray(arr)).to(dev)


def _f32(x: torch.Tensor) -> torch.Tensor:
    if not (_dry() and x.is_meta):
        _need_dev(x)
    if x.dtype != torch.float32:
        x = x.float()
    return x.contiguous()


def _dev_f(vals, dev):
    return dev_array(np.asarray(list(vals), dtype=np.float32), dev)


def _dev_i(vals, dev):
    return dev_array(np.asarray([int(v) for v in vals], dtype=np.int32), dev)


def set_selective(flag: bool) -> None:
    """Selected-sample mode (calling thread; set by BatchedAugment around one batch): a stage touches only the samples it was
    drawn for -- element-wise stages run IN PLACE, neighbourhood stages write the selected samples into a scratch buffer and
    `mia_copy_selected` puts them back, unselected samples are neither read nor written (`apply[b] = -1`, include/mia_hip.h).
    The caller's tensors are modified in place in this mode."""
    _TLS.selective = bool(flag)


def _selective() -> bool:
    return getattr(_TLS, "selective", False)


def _apply(apply, dev):
    if apply is None:
        return None
    if _selective():
        return _dev_i([1 if a else -1 for a in apply], dev)
    return _dev_i(apply, dev)


def _scratch(like: torch.Tensor) -> torch.Tensor:
    """Per-thread scratch tensor of `like`'s shape / dtype / device, reused from batch to batch (selected-sample mode)."""
    if like.is_meta:
        return torch.empty_like(like)
    pool = getattr(_TLS, "scratch", None)
    if pool is None:
        pool = _TLS.scratch = {}
    key = (tuple(like.shape), like.dtype, like.device)
    t = pool.get(key)
    if t is None:
        if len(pool) > 8:
            pool.clear()
        t = pool[key] = torch.empty_like(like)
    return t


def _sel_ok(*ts) -> bool:
    """Selected-sample mode can serve these batch tensors: 16-byte multiples per sample (mia_copy_selected)."""
    return _selective() and all(t is None or (t.numel() // t.shape[0] * t.element_size()) % 16 == 0 for t in ts)


def _copy_back(src: torch.Tensor, dst: torch.Tensor, ap) -> None:
    call("mia_copy_selected", _p(src), _p(dst), _c_i64(src.numel() // src.shape[0] * src.element_size()), src.shape[0], _p(ap), _stream())


def affine_nearest(img: Optional[torch.Tensor], lab: Optional[torch.Tensor], mats: Sequence[Sequence[float]], apply=None):
    """torchvision F.affine / F.rotate tensor path (nearest, zero fill) on image and label in one launch."""
    ref = img if img is not None else lab
    dev = ref.device
    b, h, w = ref.shape[0], ref.shape[-2], ref.shape[-1]
    c = img.shape[1] if img is not None else 1
    io = oi = lo = ol = None
    if img is not None:
        io = _f32(img)
    if lab is not None:
        if not _dry():
            _need_dev(lab)
        lo = lab.long().contiguous()
    sel = apply is not None and _sel_ok(io, lo)
    if io is not None:
        oi = _scratch(io) if sel else torch.empty_like(io)
    if lo is not None:
        ol = _scratch(lo) if sel else torch.empty_like(lo)
    m = dev_array(np.asarray([list(r) for r in mats], dtype=np.float32).reshape(b, 6), dev)
    ap = _apply(apply, dev) if (sel or not _selective()) else _dev_i(apply, dev)
    if _dry():
        return (io, lo) if sel else (oi, ol)
    call("mia_affine_nearest", _p(io), _p(oi), _p(lo), _p(ol), b, c, h, w, _p(m), _p(ap), _stream())
    if sel:  # selected samples only: scratch -> back into the batch tensors
        if io is not None:
            _copy_back(oi, io, ap)
        if lo is not None:
            _copy_back(ol, lo, ap)
        return io, lo
    return oi, ol


def elastic_warp(img: Optional[torch.Tensor], lab: Optional[torch.Tensor], disp: torch.Tensor, apply=None):
    """Elastic deformation (own spec, csrc/augment.hip): disp [B, 2, gh, gw] fp32 control-point displacements in pixels
    (component 0 = x, 1 = y); image [B, C, H, W] sampled bilinearly (zero outside), label [B, H, W] at the nearest pixel."""
    ref = img if img is not None else lab
    dev = ref.device
    b, h, w = ref.shape[0], ref.shape[-2], ref.shape[-1]
    c = img.shape[1] if img is not None else 1
    if isinstance(disp, np.ndarray):
        d = dev_array(disp.astype(np.float32), dev)
    else:
        _need_dev(disp)
        d = disp.to(device=dev, dtype=torch.float32).contiguous()
    if d.ndim != 4 or d.shape[0] != b or d.shape[1] != 2 or d.shape[2] < 2 or d.shape[3] < 2:
        raise MiaError(f"elastic_warp: displacement grid must be [B, 2, gh >= 2, gw >= 2], got {tuple(d.shape)}")
    io = oi = lo = ol = None
    if img is not None:
        io = _f32(img)
    if lab is not None:
        if not _dry():
            _need_dev(lab)
        lo = lab.long().contiguous()
    sel = apply is not None and _sel_ok(io, lo)
    if io is not None:
        oi = _scratch(io) if sel else torch.empty_like(io)
    if lo is not None:
        ol = _scratch(lo) if sel else torch.empty_like(lo)
    ap = _apply(apply, dev) if (sel or not _selective()) else _dev_i(apply, dev)
    if _dry():
        return (io, lo) if sel else (oi, ol)
    call("mia_elastic_warp", _p(io), _p(oi), _p(lo), _p(ol), b, c, h, w, _p(d), d.shape[2], d.shape[3], _p(ap), _stream())
    if sel:
        if io is not None:
            _copy_back(oi, io, ap)
        if lo is not None:
            _copy_back(ol, lo, ap)
        return io, lo
    return oi, ol


def rot90_flip(x: torch.Tensor, k: int = 0, flip_h: bool = False, flip_w: bool = False) -> torch.Tensor:
    """torch.rot90(x, k, (-2, -1)) followed by optional flips of H / W, for 4- or 8-byte dtypes."""
    if not _dry():
        _need_dev(x)
    x = x.contiguous()
    if x.element_size() not in (4, 8):
        raise MiaError("rot90_flip supports 4- or 8-byte element types")
    h, w = x.shape[-2], x.shape[-1]
    planes = x.numel() // (h * w)
    oshape = list(x.shape)
    if k % 2:
        oshape[-2], oshape[-1] = w, h
    out = torch.empty(oshape, device=x.device, dtype=x.dtype)
    if _dry():
        return out
    call("mia_rot90_flip", _p(x), _p(out), x.element_size(), 1, planes, h, w, int(k) % 4, int(flip_h), int(flip_w), _stream())
    return out


def crop(x: torch.Tensor, top: Sequence[int], left: Sequence[int], oh: int, ow: int) -> torch.Tensor:
    """x [B, ..., H, W] (4- or 8-byte dtype) -> [B, ..., oh, ow]: per-sample window x[b, ..., top[b]:top[b]+oh, left[b]:left[b]+ow]."""
    if not _dry():
        _need_dev(x)
    x = x.contiguous()
    if x.element_size() not in (4, 8):
        raise MiaError("crop supports 4- or 8-byte element types")
    b, h, w = x.shape[0], x.shape[-2], x.shape[-1]
    if len(top) != b or len(left) != b or min(top) < 0 or min(left) < 0 or max(top) + oh > h or max(left) + ow > w:
        raise MiaError(f"crop: window {oh}x{ow} at {list(top)},{list(left)} does not fit {h}x{w}")
    planes = x.numel() // (b * h * w)
    out = torch.empty(tuple(x.shape[:-2]) + (oh, ow), device=x.device, dtype=x.dtype)
    tp, lf = _dev_i(top, x.device), _dev_i(left, x.device)
    if _dry():
        return out
    call("mia_crop", _p(x), _p(out), x.element_size(), b, planes, h, w, oh, ow, _p(tp), _p(lf), _stream())
    return out


def gaussian_blur(img: torch.Tensor, sigma: Sequence[float], ksize: Sequence[int], apply=None) -> torch.Tensor:
    x = _f32(img)
    b, c, h, w = x.shape
    sel = apply is not None and _sel_ok(x)
    out = _scratch(x) if sel else torch.empty_like(x)
    sg, ks = _dev_f(sigma, x.device), _dev_i(ksize, x.device)  # keep alive across the launch
    ap = _apply(apply, x.device) if (sel or not _selective()) else _dev_i(apply, x.device)
    if _dry():
        return x if sel else out
    call("mia_gaussian_blur", _p(x), _p(out), b, c, h, w, _p(sg), _p(ks), int(max(ksize)), _p(ap), _stream())
    if sel:
        _copy_back(out, x, ap)
        return x
    return out


def sample_stats(img: torch.Tensor, gray: bool = False, apply=None) -> torch.Tensor:
    """[B, 2] = per-sample (mean, unbiased std) over C*H*W (luma image if gray and C == 3).  In selected-sample mode with an
    `apply` list only the selected samples are read (the other rows are zero)."""
    x = _f32(img)
    b, c, h, w = x.shape
    ws = torch.empty(lib().mia_sample_stats_workspace(b), device=x.device, dtype=torch.float32)
    out = torch.empty((b, 2), device=x.device, dtype=torch.float32)
    ap = _apply(apply, x.device) if (apply is not None and _selective()) else None
    if _dry():
        return out
    if ap is not None:
        call("mia_sample_stats_sel", _p(x), b, c, _c_i64(h * w), int(gray), _p(ws), _p(out), _p(ap), _stream())
    else:
        call("mia_sample_stats", _p(x), b, c, _c_i64(h * w), int(gray), _p(ws), _p(out), _stream())
    return out


def elementwise(img: torch.Tensor, op: int, p0=None, mean_std: Optional[torch.Tensor] = None,
                aux: Optional[torch.Tensor] = None, apply=None) -> torch.Tensor:
    x = _f32(img)
    b = x.shape[0]
    out = x if (_selective() and apply is not None) else torch.empty_like(x)  # selected-sample mode: in place
    pp = None if p0 is None else _dev_f(p0, x.device)
    if aux is not None:
        aux = _f32(aux)
    ap = _apply(apply, x.device)
    if _dry():
        return out
    call("mia_elementwise", _p(x), _p(out), _c_i64(x.numel() // b), b, op, _p(pp), _p(mean_std), _p(aux), _p(ap), _stream())
    return out


def noise_clip(img: torch.Tensor, sigma: Sequence[float], seed: int, offset: int = 0, apply=None) -> torch.Tensor:
    x = _f32(img)
    b = x.shape[0]
    out = x if (_selective() and apply is not None) else torch.empty_like(x)  # selected-sample mode: in place
    sg, ap = _dev_f(sigma, x.device), _apply(apply, x.device)
    if _dry():
        return out
    call("mia_noise_clip", _p(x), _p(out), _c_i64(x.numel() // b), b, _p(sg), ctypes.c_uint64(seed), ctypes.c_uint64(offset),
         _p(ap), _stream())
    return out


def resize_bilinear(img: torch.Tensor, oh: int, ow: int, antialias: bool = False) -> torch.Tensor:
    """interpolate(bilinear, align_corners=False[, antialias]) on [B, C, H, W]."""
    x = _f32(img)
    b, c, h, w = x.shape
    out = torch.empty((b, c, oh, ow), device=x.device, dtype=torch.float32)
    if _dry():
        return out
    if antialias and (oh < h or ow < w):
        tmp = torch.empty((b, c, h, ow), device=x.device, dtype=torch.float32)
        call("mia_resize_bilinear_aa", _p(x), _p(tmp), _p(out), b, c, h, w, oh, ow, _stream())
    else:
        call("mia_resize_bilinear", _p(x), _p(out), b, c, h, w, oh, ow, None, None, _stream())
    return out


def lowres(img: torch.Tensor, low_hw: Sequence[Sequence[int]], apply=None) -> torch.Tensor:
    """SimulateLowRes: nearest-exact down to low_hw[b], bilinear back up, fused."""
    x = _f32(img)
    b, c, h, w = x.shape
    sel = apply is not None and _sel_ok(x)
    out = _scratch(x) if sel else torch.empty_like(x)
    lw = dev_array(np.asarray([[int(a), int(d)] for a, d in low_hw], dtype=np.int32), x.device)
    ap = _apply(apply, x.device) if (sel or not _selective()) else _dev_i(apply, x.device)
    if _dry():
        return x if sel else out
    call("mia_resize_bilinear", _p(x), _p(out), b, c, h, w, h, w, _p(lw), _p(ap), _stream())
    if sel:
        _copy_back(out, x, ap)
        return x
    return out


def resize_nearest(x: torch.Tensor, oh: int, ow: int) -> torch.Tensor:
    """interpolate(nearest) on the last two dims; float32 images or int64 label maps."""
    if not _dry():
        _need_dev(x)
    if x.dtype not in (torch.float32, torch.int64, torch.int32):
        x = x.float() if x.is_floating_point() else x.long()
    x = x.contiguous()
    h, w = x.shape[-2], x.shape[-1]
    planes = x.numel() // (h * w)
    out = torch.empty(list(x.shape[:-2]) + [oh, ow], device=x.device, dtype=x.dtype)
    if _dry():
        return out
    call("mia_resize_nearest", _p(x), _p(out), x.element_size(), _c_i64(planes), h, w, oh, ow, _stream())
    return out


class ResizeBilinearFn(torch.autograd.Function):
    """Differentiable bilinear resize (deep-supervision ``Upsample``, reference blocks.py:45-63, unet.py:193-197)."""

    @staticmethod
    def forward(ctx, x, oh, ow):
        ctx.in_shape = tuple(x.shape)
        return resize_bilinear(x, int(oh), int(ow))

    @staticmethod
    def backward(ctx, dout):
        b, c, h, w = ctx.in_shape
        d = _f32(dout)
        din = torch.zeros((b, c, h, w), device=d.device, dtype=torch.float32)
        call("mia_resize_bilinear_bwd", _p(d), _p(din), b, c, h, w, d.shape[2], d.shape[3], _stream())
        return din, None, None
