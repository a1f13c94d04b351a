"""HostFeed.stage line by line on the GPU box (debug aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-analysis_amd")]
import torch
from training.feed import HostFeed, _fit
from mia_hip import ops
from mia_hip.ops import _c_i64, _p, call
dev = torch.device("cuda:0")
n, H0, W0 = 32, 496, 608
g = torch.Generator().manual_seed(0)
image = torch.rand(n, 1, H0, W0, generator=g); label = torch.randint(0, 3, (n, H0, W0), generator=g)
self = HostFeed(dev)
T = {}
def lap(k, t0):
    T[k] = T.get(k, 0.0) + time.perf_counter() - t0
    return time.perf_counter()
for it in range(24):
    if it == 4: T.clear()
    t = time.perf_counter()
    main = torch.cuda.current_stream(self.device)
    if self.last is not None:
        self.last.consumed = torch.cuda.Event(); self.last.consumed.record(main)
    s = self.slots[self.i % len(self.slots)]; self.i += 1
    t = lap("events", t)
    s.h2d_done.synchronize(); t = lap("h2d_done.synchronize", t)
    narrow = self.labels_fit_a_byte(label); t = lap("aminmax", t)
    s.pin_img = _fit(s.pin_img, image.shape, torch.float32, pin_memory=True); s.pin_img.copy_(image); t = lap("img->pinned", t)
    s.pin_lab = _fit(s.pin_lab, label.shape, torch.uint8, pin_memory=True); s.pin_lab.copy_(label); t = lap("lab->pinned u8", t)
    s.dev_img = _fit(s.dev_img, image.shape, torch.float32, device=self.device); s.dev_lab_raw = _fit(s.dev_lab_raw, label.shape, torch.uint8, device=self.device)
    if s.consumed is not None: self.copy_stream.wait_event(s.consumed)
    with torch.cuda.stream(self.copy_stream):
        s.dev_img.copy_(s.pin_img, non_blocking=True); s.dev_lab_raw.copy_(s.pin_lab, non_blocking=True); s.h2d_done.record(self.copy_stream)
    main.wait_event(s.h2d_done); t = lap("enqueue copies", t)
    s.dev_lab = _fit(s.dev_lab, label.shape, torch.int64, device=self.device)
    call("mia_widen_u8_i64", _p(s.dev_lab_raw), _p(s.dev_lab), _c_i64(label.numel()), ops._stream()); t = lap("widen launch", t)
    self.last = s
torch.cuda.synchronize()
print({k: round(1e3 * v / 20, 3) for k, v in T.items()})
