#!/usr/bin/env python
"""Does running a chain of PlainBlocks over batch CHUNKS (instance norm: per-image statistics, so exact) keep the
intermediates in the 256 MiB Infinity Cache and shorten the HBM-bound passes?  Times a chain of L blocks C -> C at S x S,
bf16, batch 32, forward only: whole batch layer by layer (today's schedule) vs chunk by chunk through all L layers.

    python tools/probe_batch_chunks.py [--c 64] [--size 512] [--layers 2]
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
    ap.add_argument("--c", type=int, default=64)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=7)
    a = ap.parse_args()
    from mia_hip import NORM_INSTANCE, ops
    dev = torch.device("cuda:0")
    c, s, n = a.c, a.size, a.batch
    x = torch.randn(n, s, s, c, device=dev).to(torch.bfloat16)
    ws = [torch.randn(c, c, 3, 3, device=dev) * 0.05 for _ in range(a.layers)]
    b = torch.zeros(c, device=dev)
    g, be = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    cfg = ops.NormCfg(NORM_INSTANCE, True)

    def chain(xin):
        t = xin
        for w in ws:
            t = ops.PlainBlockFn.apply(t, None, w, b, g, be, 1, cfg)
        return t

    img_mb = s * s * c * 2 / 2 ** 20
    with torch.no_grad():
        ref = chain(x)
        for k in (n, 16, 8, 4, 2, 1):
            outs = [chain(x[i:i + k]) for i in range(0, n, k)]
            same = torch.equal(torch.cat(outs), ref)
            ts = []
            for _ in range(a.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(0, n, k):
                    chain(x[i:i + k])
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            m = statistics.median(ts)
            print(f"C={c} {s}x{s} L={a.layers} chunk {k:2d} images ({k * img_mb:6.0f} MiB per tensor): {m:.3f} ms  "
                  f"({m / a.layers:.3f} per block)  bit-identical to whole batch: {same}", flush=True)


if __name__ == "__main__":
    main()
