#!/usr/bin/env python
"""Same-process A/B of one library option over every stride-1 3x3 conv launch of the cfg3 step (bf16, batch 32): forward and
input gradient at C -> C and the decoder's first conv (2C split input -> C), interleaved rounds, HIP events, medians.

    python tools/ab_levels.py conv_bt [--base 64] [--size 512] [--batch 32]
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
    ap.add_argument("--base", type=int, default=64)
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--inner", type=int, default=4)
    ap.add_argument("--on", type=int, default=1, help="value of the '1' arm (e.g. conv64_dma 2)")
    a = ap.parse_args()
    import mia_hip
    from mia_hip import BF16, CONV_G3S1, ops
    dev = torch.device("cuda:0")
    lib = mia_hip.lib()
    opt = a.option.encode()
    default = mia_hip.get_option(a.option)
    tot = {0: 0.0, 1: 0.0}
    flops = 0.0
    for lvl in range(a.levels):
        c, s = a.base << lvl, a.size >> lvl
        x = torch.randn(a.batch, s, s, c, device=dev).to(torch.bfloat16)
        x2 = torch.randn(a.batch, s, s, c, device=dev).to(torch.bfloat16)
        w = torch.randn(c, c, 3, 3, device=dev) * 0.02
        wcat = torch.randn(c, 2 * c, 3, 3, device=dev) * 0.02
        b = torch.zeros(c, device=dev)
        pc, pc2 = ops.PackCache(), ops.PackCache()
        wp, npad, kpad = pc.get(w, BF16, True)
        wb, npb, kpb = pc.get(w, BF16, False)
        wc, npc, kpc = pc2.get(wcat, BF16, True)
        wcb, npcb, kpcb = pc2.get(wcat, BF16, False)
        runs = {
            "fwd   C->C": (lambda: ops.conv_mma(CONV_G3S1, x, None, wp, npad, kpad, False, b, c, (s, s), want_stats=True), c * c),
            "dgrad C->C": (lambda: ops.conv_mma(CONV_G3S1, x, None, wb, npb, kpb, True, None, c, (s, s)), c * c),
            "fwd  2C->C": (lambda: ops.conv_mma(CONV_G3S1, x, x2, wc, npc, kpc, False, b, c, (s, s), want_stats=True), 2 * c * c),
            "dgrad C->2C": (lambda: ops.conv_mma(CONV_G3S1, x, None, wcb, npcb, kpcb, True, None, 2 * c, (s, s), out_split=c), 2 * c * c),
        }
        for name, (fn, cc) in runs.items():
            if lvl == a.levels - 1 and "2C" in name:
                continue
            times = {0: [], 1: []}
            for r in range(a.rounds):
                for flag in ((0, 1) if r % 2 else (1, 0)):
                    lib.mia_set_option(opt, a.on if flag else 0)
                    fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.inner):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    times[flag].append(e0.elapsed_time(e1) / a.inner)
            fl = 2.0 * 9 * cc * s * s * a.batch
            m0, m1 = statistics.median(times[0]), statistics.median(times[1])
            tot[0] += m0
            tot[1] += m1
            flops += fl
            print(f"level {lvl} {name:11s} C={c:4d} {s:3d}x{s:3d}: {a.option}=0 {m0:.3f} ms {fl / m0 / 1e9:7.1f} TF | =1 {m1:.3f} ms {fl / m1 / 1e9:7.1f} TF | ratio {m1 / m0:.3f}",
                  flush=True)
    lib.mia_set_option(opt, default)
    print(f"sum: {a.option}=0 {tot[0]:.3f} ms ({flops / tot[0] / 1e9:.0f} TF) | =1 {tot[1]:.3f} ms ({flops / tot[1] / 1e9:.0f} TF) | ratio {tot[1] / tot[0]:.3f}")


if __name__ == "__main__":
    main()
