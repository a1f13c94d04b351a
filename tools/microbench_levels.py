#!/usr/bin/env python
"""Per-level timing of the stride-1 3x3 conv launches of the cfg3 step (bf16, batch 32): forward and dgrad at C -> C and the
decoder's first conv (2C split input -> C).  MB_DTYPE=f32 MB_SIZE=256: the same launches of cfg2 (fp32 tensors, split-f16 products)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "medical-image-analysis_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    import mia_hip
    from mia_hip import BF16, F32, CONV_G3S1, ops
    f32 = os.environ.get("MB_DTYPE", "bf16") == "f32"
    DT, tdt, size0 = (F32, torch.float32, int(os.environ.get("MB_SIZE", "256"))) if f32 else (BF16, torch.bfloat16, int(os.environ.get("MB_SIZE", "512")))
    dev = torch.device("cuda:0")
    B, iters = int(os.environ.get("MB_BATCH", "32")), int(os.environ.get("MB_ITERS", "10"))
    chans = [64, 128, 256, 512, 1024]
    tot = 0.0
    for lvl in range(5):
        c, s = chans[lvl], size0 >> lvl
        x = torch.randn(B, s, s, c, device=dev).to(tdt)
        x2 = torch.randn(B, s, s, c, device=dev).to(tdt)
        w = torch.randn(c, c, 3, 3, device=dev) * 0.02
        wcat = torch.randn(c, 2 * c, 3, 3, device=dev) * 0.02
        b = torch.zeros(c, device=dev)
        pc, pc2 = ops.PackCache(), ops.PackCache()
        wp, npad, kpad = pc.get(w, DT, True)
        wb, npb, kpb = pc.get(w, DT, False)
        wc, npc, kpc = pc2.get(wcat, DT, True)
        runs = {
            "fwd   C->C": (lambda: ops.conv_mma(CONV_G3S1, x, None, wp, npad, kpad, False, b, c, (s, s), want_stats=True), c * c),
            "dgrad C->C": (lambda: ops.conv_mma(CONV_G3S1, x, None, wb, npb, kpb, True, None, c, (s, s)), c * c),
            "fwd  2C->C": (lambda: ops.conv_mma(CONV_G3S1, x, x2, wc, npc, kpc, False, b, c, (s, s), want_stats=True), 2 * c * c),
        }
        for name, (fn, cc) in runs.items():
            if lvl == 4 and name.startswith("fwd  2C"):
                continue
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / iters * 1e3
            tot += ms
            print(f"level {lvl} {name} C={c:4d} {s:3d}x{s:3d}: {ms:.3f} ms  {2.0 * 9 * cc * s * s * B / ms / 1e9:7.1f} TFLOP/s")
    print(f"sum {tot:.3f} ms")


if __name__ == "__main__":
    main()
