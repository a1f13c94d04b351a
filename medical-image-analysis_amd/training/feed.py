"""Host -> device feed of the train step (reference: `image.to(device)`, `label.to(device)` on DataLoader output,
al_trainer.py:1366-1368, fugc_dataset.py:140-164).

The reference's copies come from pageable memory and therefore block the host for the whole transfer (33 MB of fp32 image + 67 MB of
int64 labels per cfg3 step).  `HostFeed` keeps a small ring of pinned staging buffers and device buffers: `stage()` copies the batch
into pinned memory (the labels as uint8 when every value fits in a byte -- checked, never assumed), enqueues the two H2D copies on
a side stream and makes the launch stream wait for them; the labels are widened to int64 on the device (`mia_widen_u8_i64`).  The
host returns at once, so the copies of step i + 1 overlap the kernels of step i.  `TrainEngine.train_step` does this by itself for
batches that arrive in host memory; `DevicePrefetcher` wraps any iterable of batches for loops that want the device tensors."""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, Optional, Tuple

import ctypes

import torch

from mia_hip import lib, ops
from mia_hip.ops import _c_i64, _p, call


class _Slot:
    __slots__ = ("pin_img", "pin_lab", "pin_lab8", "pin_lab64", "dev_img", "dev_lab_raw", "dev_lab", "h2d_done", "consumed")

    def __init__(self):
        self.pin_img = self.pin_lab = self.pin_lab8 = self.pin_lab64 = self.dev_img = self.dev_lab_raw = self.dev_lab = None
        self.h2d_done = torch.cuda.Event()
        self.consumed: Optional[torch.cuda.Event] = None


def _fit(buf: Optional[torch.Tensor], shape, dtype, **kw) -> torch.Tensor:
    if buf is None or buf.shape != torch.Size(shape) or buf.dtype != dtype:
        return torch.empty(shape, dtype=dtype, **kw)
    return buf


class HostFeed:
    RING = 3

    def __init__(self, device, ring: int = RING):
        self.device = torch.device(device)
        self.slots = [_Slot() for _ in range(ring)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.i = 0
        self.last: Optional[_Slot] = None
        self.bytes_h2d = 0  # of the last staged batch (bench.py reports it)

    @staticmethod
    def labels_fit_a_byte(label: torch.Tensor) -> bool:
        """True when every int64 label lies in 0 .. 255 (reference implementation of the check `stage` performs inside
        `mia_host_narrow_labels`; tests compare the two)."""
        if label.dtype != torch.int64 or label.numel() == 0:
            return False
        lo, hi = torch.aminmax(label)
        return int(lo) >= 0 and int(hi) <= 255

    def stage(self, image: torch.Tensor, label: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(image fp32, label int64) on the device for a batch in host memory; stream-ordered on the current stream."""
        main = torch.cuda.current_stream(self.device)
        if self.last is not None:  # the step that consumed the previous batch has been enqueued by now: its buffers are free after it
            self.last.consumed = torch.cuda.Event()
            self.last.consumed.record(main)
        s = self.slots[self.i % len(self.slots)]
        self.i += 1
        s.h2d_done.synchronize()  # this slot's pinned buffers: the copies that last read them are done (the host may run ahead)
        image = image if image.dtype == torch.float32 else image.float()
        label = label if label.is_contiguous() else label.contiguous()
        image = image if image.is_contiguous() else image.contiguous()
        s.pin_img = _fit(s.pin_img, image.shape, torch.float32, pin_memory=True)
        # (plain threads inside the library, not torch's OpenMP pool: see mia_host_copy)
        lib().mia_host_copy(ctypes.c_void_p(s.pin_img.data_ptr()), ctypes.c_void_p(image.data_ptr()), ctypes.c_int64(image.numel() * 4), 0)
        # labels: narrowed into the pinned byte buffer and range-checked in ONE pass (mia_host_narrow_labels); a label outside
        # 0 .. 255 sends the int64 tensor instead
        s.pin_lab8 = _fit(s.pin_lab8, label.shape, torch.uint8, pin_memory=True)
        narrow = lib().mia_host_narrow_labels(ctypes.c_void_p(label.data_ptr()), ctypes.c_void_p(s.pin_lab8.data_ptr()),
                                              ctypes.c_int64(label.numel()), 0) == 1
        if narrow:
            s.pin_lab = s.pin_lab8
        else:
            s.pin_lab64 = _fit(s.pin_lab64, label.shape, torch.int64, pin_memory=True)
            lib().mia_host_copy(ctypes.c_void_p(s.pin_lab64.data_ptr()), ctypes.c_void_p(label.data_ptr()), ctypes.c_int64(label.numel() * 8), 0)
            s.pin_lab = s.pin_lab64
        lab_dt = s.pin_lab.dtype
        fresh = (s.dev_img is None or s.dev_img.shape != image.shape or s.dev_lab_raw is None or s.dev_lab_raw.shape != label.shape
                 or s.dev_lab_raw.dtype != lab_dt)
        s.dev_img = _fit(s.dev_img, image.shape, torch.float32, device=self.device)
        s.dev_lab_raw = _fit(s.dev_lab_raw, label.shape, lab_dt, device=self.device)
        if fresh:
            # A block the caching allocator has just handed out may still be in use by launches that are IN FLIGHT on the launch
            # stream (it re-uses memory stream-ordered: freed on the host, not yet on the device).  Writing it from the side stream
            # before those launches have drained overwrote live activations of the previous step -- NaN from the second host-fed
            # step on (round 5; the small bit-identity test did not hit it, cfg3 did).  The copy stream therefore waits for the
            # launch stream once per new buffer.
            ev = torch.cuda.Event()
            ev.record(main)
            self.copy_stream.wait_event(ev)
        if s.consumed is not None:
            self.copy_stream.wait_event(s.consumed)  # the step that used this slot's device buffers has finished with them
        with torch.cuda.stream(self.copy_stream):
            s.dev_img.copy_(s.pin_img, non_blocking=True)
            s.dev_lab_raw.copy_(s.pin_lab, non_blocking=True)
            s.h2d_done.record(self.copy_stream)
        main.wait_event(s.h2d_done)
        self.bytes_h2d = s.pin_img.numel() * 4 + s.pin_lab.numel() * s.pin_lab.element_size()
        if narrow:
            s.dev_lab = _fit(s.dev_lab, label.shape, torch.int64, device=self.device)
            call("mia_widen_u8_i64", _p(s.dev_lab_raw), _p(s.dev_lab), _c_i64(label.numel()), ops._stream())
            lab = s.dev_lab
        else:
            lab = s.dev_lab_raw
        self.last = s
        return s.dev_img, lab


class DevicePrefetcher:
    """Iterate over `batches` ({"image", "label"} dicts in host memory, e.g. a DataLoader) and yield the same dicts with device
    tensors, staged through a `HostFeed`: the H2D copies of a batch are in flight while the previous one trains.  A yielded batch
    stays valid until `ring - 1` further batches have been drawn."""

    def __init__(self, batches: Iterable[Dict[str, torch.Tensor]], device, ring: int = HostFeed.RING):
        self.batches, self.feed = batches, HostFeed(device, ring)

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        for b in self.batches:
            img, lab = self.feed.stage(b["image"], b["label"])
            out = dict(b)
            out["image"], out["label"] = img, lab
            yield out
