#!/usr/bin/env python
"""Per-level timing of the weight-gradient launches of the cfg3 step (bf16, batch 32), launch + slab reduce."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "medical-image-analysis_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    from mia_hip import WGRAD_2S2, WGRAD_3S1, WGRAD_3S2, ops
    dev = torch.device("cuda:0")
    B, iters = int(os.environ.get("MB_BATCH", "32")), int(os.environ.get("MB_ITERS", "10"))
    chans = [64, 128, 256, 512, 1024]
    tot = 0.0
    for lvl in range(5):
        c, s = chans[lvl], 512 >> lvl
        x = torch.randn(B, s, s, c, device=dev).to(torch.bfloat16)
        x2 = torch.randn(B, s, s, c, device=dev).to(torch.bfloat16)
        dy = torch.randn(B, s, s, c, device=dev).to(torch.bfloat16)
        runs = {"3x3 s1  C->C": (lambda: ops.conv_wgrad(WGRAD_3S1, x, None, dy, (c, c, 3, 3), c, c), 9 * c * c * s * s),
                "3x3 s1 2C->C": (lambda: ops.conv_wgrad(WGRAD_3S1, x, x2, dy, (c, 2 * c, 3, 3), c, 2 * c), 18 * c * c * s * s)}
        if lvl < 4:
            c2 = chans[lvl + 1]
            dyc = torch.randn(B, s // 2, s // 2, c2, device=dev).to(torch.bfloat16)
            runs["3x3 s2 C->2C"] = (lambda: ops.conv_wgrad(WGRAD_3S2, x, None, dyc, (c2, c, 3, 3), c2, c), 9 * c * c2 * (s // 2) ** 2)
            runs["convT 2C->C "] = (lambda: ops.conv_wgrad(WGRAD_2S2, dy, None, dyc, (c2, c, 2, 2), c2, c), 4 * c * c2 * (s // 2) ** 2)
        for name, (fn, macs) in runs.items():
            if lvl == 4 and name.startswith("3x3 s1 2C"):
                continue
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / iters * 1e3
            tot += ms
            print(f"level {lvl} wgrad {name} C={c:4d} {s:3d}x{s:3d}: {ms:.3f} ms  {2.0 * macs * B / ms / 1e9:7.1f} TFLOP/s")
    print(f"sum {tot:.3f} ms")


if __name__ == "__main__":
    main()
