#!/usr/bin/env python
"""Same-process A/B of one library option (mia_set_option) on one op: interleaved rounds, HIP events, median / min per arm.

    python tools/ab_option.py wgrad_tab wgrad --c 64 --size 512
    python tools/ab_option.py conv64 conv --c 64 --size 512
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "medical-image-analysis_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("option")
    ap.add_argument("what", choices=["conv", "dgrad", "wgrad"])
    ap.add_argument("--c", type=int, default=64)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--inner", type=int, default=5)
    a = ap.parse_args()
    import mia_hip
    from mia_hip import CONV_G3S1, WGRAD_3S1, ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(a.batch, a.size, a.size, a.c, generator=g).to(dev).to(torch.bfloat16)
    dy = torch.randn(a.batch, a.size, a.size, a.c, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(a.c, a.c, 3, 3, generator=g) / (3 * a.c ** 0.5)).to(dev)
    b = torch.randn(a.c, generator=g).to(dev)
    pc = ops.PackCache()
    lib = mia_hip.lib()

    def op():
        if a.what == "wgrad":
            return ops.conv_wgrad(WGRAD_3S1, x, None, dy, w.shape, a.c, a.c)
        flip = a.what == "dgrad"
        wp, npad, kpad = pc.get(w, mia_hip.BF16, not flip)  # square weights: either packing has the right shape for timing
        return ops.conv_mma(CONV_G3S1, x, None, wp, npad, kpad, flip, None if flip else b, a.c, (a.size, a.size), want_stats=not flip)[0]

    outs = []
    for flag in (1, 0):
        lib.mia_set_option(a.option.encode(), flag)
        outs.append(op().float().clone())
    torch.cuda.synchronize()
    print(f"{a.option}: max |on - off| = {(outs[0] - outs[1]).abs().max().item():.3e}  (|off| max {outs[1].abs().max().item():.3e})")
    times = {0: [], 1: []}
    for r in range(a.rounds):
        for flag in ((0, 1) if r % 2 else (1, 0)):
            lib.mia_set_option(a.option.encode(), flag)
            op()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.inner):
                op()
            e1.record()
            torch.cuda.synchronize()
            times[flag].append(e0.elapsed_time(e1) / a.inner)
    for flag in (0, 1):
        t = times[flag]
        print(f"{a.what} c={a.c} {a.size}x{a.size} {a.option}={flag}: median {statistics.median(t):.4f} ms  min {min(t):.4f} ms")
    lib.mia_set_option(a.option.encode(), 1)


if __name__ == "__main__":
    main()
