#!/usr/bin/env python
"""Per-level timing of every stride-2 launch of the cfg3 step (bf16, batch 32): the strided 3x3 conv (forward, input gradient,
weight gradient) and ConvTranspose2d 2x2 (forward, input gradient, weight gradient), HIP events, medians, with the two
floors beside each: MFMA (2.5 PFLOP/s) and HBM (algorithmic bytes at 8 TB/s).

    python tools/s2_levels.py [--base 64] [--size 512] [--batch 32] [--option NAME --values 0,1]
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
    ap.add_argument("--base", type=int, default=64)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--inner", type=int, default=4)
    ap.add_argument("--option", default=None)
    ap.add_argument("--values", default="0,1")
    a = ap.parse_args()
    import mia_hip
    from mia_hip import BF16, CONV_G2S2, CONV_G3S2, CONV_T2S2, CONV_T3S2, WGRAD_2S2, WGRAD_3S2, ops
    dev = torch.device("cuda:0")
    values = [int(v) for v in a.values.split(",")] if a.option else [None]
    default = mia_hip.get_option(a.option) if a.option else None
    tot = {v: 0.0 for v in values}
    for lvl in range(a.levels):
        c, s = a.base << lvl, a.size >> lvl  # level lvl: c channels at s x s; level lvl + 1: 2c channels at s/2 x s/2
        n, h = a.batch, s // 2
        fine = torch.randn(n, s, s, c, device=dev).to(torch.bfloat16)
        coarse = torch.randn(n, h, h, 2 * c, device=dev).to(torch.bfloat16)
        w3 = torch.randn(2 * c, c, 3, 3, device=dev) * 0.02          # Conv2d(c -> 2c, 3, stride 2)
        wt = torch.randn(2 * c, c, 2, 2, device=dev) * 0.02          # ConvTranspose2d(2c -> c, 2, 2)
        b2, b1 = torch.zeros(2 * c, device=dev), torch.zeros(c, device=dev)
        pc3, pct = ops.PackCache(), ops.PackCache()
        w3f, n3f, k3f = pc3.get(w3, BF16, True)
        w3b, n3b, k3b = pc3.get(w3, BF16, False)
        wtf, ntf, ktf = pct.get(wt, BF16, False)
        wtb, ntb, ktb = pct.get(wt, BF16, True)
        px_f, px_c = n * s * s, n * h * h
        by_f, by_c = px_f * c * 2, px_c * 2 * c * 2
        runs = {
            "conv s2 fwd  ": (lambda: ops.conv_mma(CONV_G3S2, fine, None, w3f, n3f, k3f, False, b2, 2 * c, (h, h), want_stats=True),
                              2.0 * 9 * c * 2 * c * px_c, by_f + by_c),
            "conv s2 dgrad": (lambda: ops.conv_mma(CONV_T3S2, coarse, None, w3b, n3b, k3b, False, None, c, (s, s)),
                              2.0 * 9 * c * 2 * c * px_c, by_f + by_c),
            "conv s2 wgrad": (lambda: ops.conv_wgrad(WGRAD_3S2, fine, None, coarse, w3.shape, 2 * c, c),
                              2.0 * 9 * c * 2 * c * px_c, by_f + by_c),
            "convT fwd    ": (lambda: ops.conv_mma(CONV_T2S2, coarse, None, wtf, ntf, ktf, False, b1, c, (s, s)),
                              2.0 * 4 * c * 2 * c * px_c, by_f + by_c),
            "convT dgrad  ": (lambda: ops.conv_mma(CONV_G2S2, fine, None, wtb, ntb, ktb, False, None, 2 * c, (h, h)),
                              2.0 * 4 * c * 2 * c * px_c, by_f + by_c),
            "convT wgrad  ": (lambda: ops.conv_wgrad(WGRAD_2S2, fine, None, coarse, wt.shape, 2 * c, c),
                              2.0 * 4 * c * 2 * c * px_c, by_f + by_c),
        }
        for name, (fn, fl, by) in runs.items():
            times = {v: [] for v in values}
            for r in range(a.rounds):
                for v in (values if r % 2 else values[::-1]):
                    if a.option:
                        mia_hip.set_option(a.option, v)
                    fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.inner):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    times[v].append(e0.elapsed_time(e1) / a.inner)
            cells = []
            for v in values:
                m = statistics.median(times[v])
                tot[v] += m
                cells.append(f"{'' if v is None else f'{a.option}={v} '}{m:.3f} ms {fl / m / 1e9:6.0f} TF {by / m / 1e6:5.0f} GB/s")
            print(f"level {lvl}->{lvl + 1} {name} C={c:4d}->{2 * c:4d} {s:3d}->{h:3d}: " + " | ".join(cells) +
                  f" | floors: mfma {fl / 2.5e12:.3f} ms, hbm {by / 8e9:.3f} ms", flush=True)
    if a.option:
        mia_hip.set_option(a.option, default)
    print("sum: " + " | ".join(f"{'' if v is None else f'{a.option}={v} '}{tot[v]:.3f} ms" for v in values))


if __name__ == "__main__":
    main()
