#!/usr/bin/env python
"""Do an MFMA-bound weight-gradient launch and an HBM-bound norm/activation stream share the chip when issued on two HIP
streams?  Times, per level of the cfg3 step (bf16, batch 32): the weight gradient alone, R passes of mia_norm_act_fwd alone,
both back to back on one stream, and both on two streams.

    python tools/probe_overlap.py [--levels 3] [--passes 2]
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
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=9)
    a = ap.parse_args()
    from mia_hip import BF16, CONV_G3S1, WGRAD_3S1, call, ops
    from mia_hip.ops import _c_float, _c_i64, _p
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream()
    for lvl in range(a.levels):
        c, s, n = a.base << lvl, a.size >> lvl, a.batch
        x = torch.randn(n, s, s, c, device=dev).to(torch.bfloat16)
        dy = torch.randn(n, s, s, c, device=dev).to(torch.bfloat16)
        y = torch.randn(n, s, s, c, device=dev).to(torch.bfloat16)
        z = torch.empty_like(y)
        w = torch.randn(c, c, 3, 3, device=dev) * 0.02
        wb, npb, kpb = ops.PackCache().get(w, BF16, False)
        sc, sh = torch.ones(n, c, device=dev), torch.zeros(n, c, device=dev)

        def mfma_job(kind):
            if kind == "wgrad":
                return ops.conv_wgrad(WGRAD_3S1, x, None, dy, w.shape, c, c)
            return ops.conv_mma(CONV_G3S1, dy, None, wb, npb, kpb, True, None, c, (s, s))

        def hbm_job(stream):
            for _ in range(a.passes):
                call("mia_norm_act_fwd", _p(y), _p(z), BF16, _p(sc), _p(sh), n, _c_i64(s * s), c, _c_float(0.01), None, stream.cuda_stream)

        main_s = torch.cuda.current_stream()

        def timed(fn):
            ts = []
            for _ in range(a.rounds):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            return statistics.median(ts)

        for kind in ("wgrad", "dgrad"):
            def both_two_streams():
                side.wait_stream(main_s)
                with torch.cuda.stream(side):
                    mfma_job(kind)
                hbm_job(main_s)
                main_s.wait_stream(side)

            mfma_job(kind); hbm_job(main_s); both_two_streams()
            t_m = timed(lambda: mfma_job(kind))
            t_h = timed(lambda: hbm_job(main_s))
            t_seq = timed(lambda: (mfma_job(kind), hbm_job(main_s)))
            t_par = timed(both_two_streams)
            print(f"level {lvl} C={c:4d} {s:3d}x{s:3d}: {kind} {t_m:.3f} ms | {a.passes} norm passes {t_h:.3f} ms | one stream {t_seq:.3f} ms | "
                  f"two streams {t_par:.3f} ms ({100 * (t_seq - t_par) / t_seq:+.1f} % saved)", flush=True)


if __name__ == "__main__":
    main()
