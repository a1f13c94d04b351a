"""f32_split = 2 (three-way split): accuracy against fp64 math and the exact fp32 kernel, and timing, per cfg2 level (conv only so far)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-analysis_amd")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import mia_hip  # noqa: E402
from mia_hip import CONV_G3S1, CONV_G3S2, CONV_T3S2, ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
# accuracy
n, c, co, h, w = 2, 64, 96, 33, 47
x = torch.randn(n, c, h, w, generator=g)
wt = torch.randn(co, c, 3, 3, generator=g) / (c * 9) ** 0.5
b = torch.randn(co, generator=g)
want = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
for v in (0, 1, 2):
    mia_hip.set_option("f32_split", v)
    wp, npad, kpad = ops.PackCache().get(wt.to(dev), mia_hip.F32, True)
    y, _, _ = ops.conv_mma(CONV_G3S1, xd, None, wp, npad, kpad, False, b.to(dev), co, (h, w))
    e = (y.cpu().permute(0, 3, 1, 2).double() - want).abs().max().item() / want.abs().max().item()
    print(f"f32_split={v}: max-norm relative error vs fp64 {e:.2e}")
# timing per level
for cc, s in ((64, 256), (128, 128), (256, 64), (512, 32), (1024, 16)):
    xx = torch.randn(32, s, s, cc, device=dev)
    ww = torch.randn(cc, cc, 3, 3, device=dev) / (cc * 9) ** 0.5
    bb = torch.zeros(cc, device=dev)
    line = f"conv {cc}->{cc} @{s}^2 x32:"
    for v in (0, 1, 2):
        mia_hip.set_option("f32_split", v)
        wp, npad, kpad = ops.PackCache().get(ww, mia_hip.F32, True)
        for _ in range(3):
            ops.conv_mma(CONV_G3S1, xx, None, wp, npad, kpad, False, bb, cc, (s, s), want_stats=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv_mma(CONV_G3S1, xx, None, wp, npad, kpad, False, bb, cc, (s, s), want_stats=True)
        e1.record()
        torch.cuda.synchronize()
        line += f"  split={v} {e0.elapsed_time(e1) / 10 * 1e3:6.0f} us"
    print(line, flush=True)
mia_hip.set_option("f32_split", 0)
