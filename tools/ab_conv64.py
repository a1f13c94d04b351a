#!/usr/bin/env python
"""A/B of the 64 -> 64 channel conv launch in ONE process (interleaved rounds, HIP events): persistent register-weight
kernel (conv64.hip) vs the generic tile kernel (conv_mma_fast.hip).  Prints per-arm median / min and checks the outputs
against each other first."""
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "medical-image-analysis_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    import mia_hip
    from mia_hip import CONV_G3S1, ops
    size, batch, rounds = int(os.environ.get("AB_SIZE", "512")), int(os.environ.get("AB_BATCH", "32")), int(os.environ.get("AB_ROUNDS", "12"))
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(batch, size, size, 64, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev)
    b = torch.randn(64, generator=g).to(dev)
    pc = ops.PackCache()
    wp, npad, kpad = pc.get(w, mia_hip.BF16, True)
    lib = mia_hip.lib()

    def run(flag, flip=False, stats=True):
        lib.mia_set_option(b"conv64", flag)
        return ops.conv_mma(CONV_G3S1, x, None, wp, npad, kpad, flip, None if flip else b, 64, (size, size), want_stats=stats)

    for flip in (False, True):
        y1, _, s1 = run(1, flip)
        y0, _, s0 = run(0, flip)
        torch.cuda.synchronize()
        d = (y1.float() - y0.float()).abs().max().item()
        ds = ((s1.sum(1) - s0.sum(1)).abs().max() / s0.sum(1).abs().max()).item()
        print(f"flip={flip}: max |y_new - y_old| = {d:.3e} (bf16 outputs, |y| max {y0.float().abs().max().item():.2f}), stats rel diff {ds:.2e}")
    times = {0: [], 1: []}
    for r in range(rounds + 2):
        for flag in (1, 0):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                run(flag)
            e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                times[flag].append(e0.elapsed_time(e1) / 4)
    fl = 2.0 * 9 * 64 * 64 * size * size * batch
    by = 2.0 * 64 * size * size * batch * 2 + 9 * 64 * 64 * 2
    for flag, name in ((1, "conv64 persistent"), (0, "generic tile kernel")):
        med, mn = statistics.median(times[flag]), min(times[flag])
        print(f"{name:22s}: median {med:.4f} ms  min {mn:.4f} ms   {fl / med / 1e9:.0f} TFLOP/s  {by / med / 1e6:.0f} GB/s ({by / med / 1e6 / 8000:.3f} of 8 TB/s)")


if __name__ == "__main__":
    main()
