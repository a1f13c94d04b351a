"""Stride-2 input gradient (T3S2) per cfg3 level: tile kernel vs conv_pw ring (option conv_pw_t3), plain and accumulating.
    python tools/probe/t3_levels.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-analysis_amd")]
import torch  # noqa: E402

import mia_hip  # noqa: E402
from mia_hip import CONV_T3S2, call, ops  # noqa: E402
from mia_hip.ops import _p, _stream  # noqa: E402

dev = torch.device("cuda:0")
n = 32
for cout, cin, hc in ((128, 64, 256), (256, 128, 128), (512, 256, 64), (1024, 512, 32)):
    dy = torch.randn(n, hc, hc, cout, device=dev).to(torch.bfloat16)
    wt = torch.randn(cout, cin, 3, 3, device=dev) / (cout * 2.25) ** 0.5
    wb, npad, kpad = ops.PackCache().get(wt, mia_hip.BF16, False)
    acc = torch.zeros(n, 2 * hc, 2 * hc, cin, device=dev, dtype=torch.bfloat16)
    line = f"dy {cout}ch @{hc}^2 -> dx {cin}ch @{2 * hc}^2:"
    for v in (0, 1):
        mia_hip.set_option("conv_pw_t3", v)
        for mode in ("plain", "acc"):
            def run():
                if mode == "plain":
                    ops.conv_mma(CONV_T3S2, dy, None, wb, npad, kpad, False, None, cin, (2 * hc, 2 * hc))
                else:
                    call("mia_conv_mma_acc", CONV_T3S2, mia_hip.BF16, _p(dy), cout, _p(wb), npad, kpad, 0, _p(acc), cin, n, hc, hc, 2 * hc, 2 * hc, None, None, None, _stream())
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            line += f"  {'ring' if v else 'tile'} {mode} {e0.elapsed_time(e1) / 20 * 1e3:6.0f} us"
    print(line, flush=True)
mia_hip.set_option("conv_pw_t3", 1)
