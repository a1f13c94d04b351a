#!/usr/bin/env python
"""Randomised A/B of every kernel-selection option against the kernel it replaces, on random (odd) shapes: the round-3 kernels
(conv_bt, conv_pw, conv_s2_wide, conv64_dma, wgrad_bt) must agree with the tile kernels bit for bit where the accumulation
order is the same (ConvTranspose forward, 512-thread stride-2 forward) and to one bf16 ulp / fp32 summation order elsewhere.  Prints one line per case; exits 1 on a mismatch.

    python tools/fuzz_kernels.py [--cases 60] [--seed 0]
"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "medical-image-analysis_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


def relerr(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-30)).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    import mia_hip
    from mia_hip import BF16, CONV_G2S2, CONV_G3S1, CONV_G3S2, CONV_T2S2, WGRAD_2S2, WGRAD_3S1, WGRAD_3S2, ops
    dev = torch.device("cuda:0")
    rng = random.Random(a.seed)
    bad = 0
    bf = torch.bfloat16

    def t(*shape):
        return torch.randn(*shape, device=dev).to(bf)

    def ab(option, on, fn):
        default = mia_hip.get_option(option)
        try:
            mia_hip.set_option(option, on)
            x1 = fn()
            mia_hip.set_option(option, 0)
            x0 = fn()
        finally:
            mia_hip.set_option(option, default)
        torch.cuda.synchronize()
        return x1, x0

    for case in range(a.cases):
        kind = rng.choice(["conv_bt", "conv_pw", "conv_s2_wide", "conv64_dma", "wgrad_bt", "wgrad_bt_s2", "conv_pw_s2", "wgrad_t2", "f32_split",
                           "f32_split"])
        n = rng.choice([1, 2, 3])
        h, w = rng.randint(3, 70), rng.randint(3, 90)
        msg, ok = "", True
        if kind == "conv_bt":
            c1 = rng.choice([64, 96, 128, 192, 256])
            two = rng.random() < 0.3
            cout = rng.choice([64, 96, 128, 192, 256, 384])
            h = max(h, 9)
            x1, x2 = t(n, h, w, c1), (t(n, h, w, c1) if two else None)
            wt = torch.randn(cout, c1 * (2 if two else 1), 3, 3, device=dev) * 0.03
            b = torch.randn(cout, device=dev)
            flip = rng.random() < 0.5
            wp, npad, kpad = ops.PackCache().get(wt, BF16, True)
            (y1, _, s1), (y0, _, s0) = ab("conv_bt", 1, lambda: ops.conv_mma(CONV_G3S1, x1, x2, wp, npad, kpad, flip, b, cout, (h, w), want_stats=True))
            # (the big tile sums the taps column-major: one bf16 ulp where the fp32 sums round apart)
            ok = relerr(y1, y0) <= 2 ** -7 and torch.allclose(s1.sum(1), s0.sum(1), rtol=1e-3, atol=1e-3 * s0.sum(1).abs().max().item())
            msg = f"c1={c1} two={two} cout={cout} flip={flip} relerr {relerr(y1, y0):.1e}"
        elif kind == "conv_pw":
            cin, cout = rng.choice([(128, 64), (256, 128), (512, 256), (192, 64), (384, 192)])
            x, dy = t(n, h, w, cin), t(n, 2 * h, 2 * w, cout)
            wt = torch.randn(cin, cout, 2, 2, device=dev) * 0.05
            b = torch.randn(cout, device=dev)
            pc = ops.PackCache()
            wf, nf, kf = pc.get(wt, BF16, False)
            wb, nb, kb = pc.get(wt, BF16, True)
            (y1, d1), (y0, d0) = ab("conv_pw", 1, lambda: (ops.conv_mma(CONV_T2S2, x, None, wf, nf, kf, False, b, cout, (2 * h, 2 * w))[0],
                                                          ops.conv_mma(CONV_G2S2, dy, None, wb, nb, kb, False, None, cin, (h, w))[0]))
            ok = torch.equal(y1, y0) and relerr(d1, d0) < 1e-2
            msg = f"cin={cin} cout={cout} dgrad relerr {relerr(d1, d0):.1e}"
        elif kind in ("conv_s2_wide", "conv_pw_s2"):
            cin, cout = rng.choice([(64, 128), (128, 256), (256, 512), (128, 128), (32, 128)])
            if kind == "conv_pw_s2":
                cin = max(cin, 64)
            hh, ww = 2 * h - rng.randint(0, 1), 2 * w - rng.randint(0, 1)
            x = t(n, hh, ww, cin)
            wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.03
            b = torch.randn(cout, device=dev)
            wp, npad, kpad = ops.PackCache().get(wt, BF16, True)
            ho, wo = (hh + 1) // 2, (ww + 1) // 2
            if kind == "conv_s2_wide":
                (y1, _, s1), (y0, _, s0) = ab("conv_s2_wide", 2, lambda: ops.conv_mma(CONV_G3S2, x, None, wp, npad, kpad, False, b, cout, (ho, wo), want_stats=True))
                ok = torch.equal(y1, y0) and torch.allclose(s1.sum(1), s0.sum(1), rtol=1e-3, atol=1e-1)
            else:
                st = rng.random() < 0.5
                (y1, _, s1), (y0, _, s0) = ab("conv_pw_s2", 2, lambda: ops.conv_mma(CONV_G3S2, x, None, wp, npad, kpad, False, b, cout, (ho, wo), want_stats=st))
                ok = relerr(y1, y0) < 1e-2 and (not st or torch.allclose(s1.sum(1), s0.sum(1), rtol=1e-3, atol=1e-1))
            msg = f"cin={cin} cout={cout} in {hh}x{ww}"
        elif kind == "conv64_dma":
            h = max(h, 9)
            x = t(n, h, w, 64)
            two = rng.random() < 0.5
            flip = two or rng.random() < 0.5
            wt = torch.randn(64, 128 if two else 64, 3, 3, device=dev) * 0.04
            wp, npad, kpad = ops.PackCache().get(wt, BF16, not two)
            b = None if flip else torch.randn(64, device=dev)
            fn = lambda: ops.conv_mma(CONV_G3S1, x, None, wp, npad, kpad, flip, b, 128 if two else 64, (h, w), want_stats=not flip,  # noqa: E731
                                      out_split=64 if two else None)
            r1, r0 = ab("conv64_dma", 2, fn)
            ok = relerr(r1[0], r0[0]) < 1e-2 and (r1[1] is None or relerr(r1[1], r0[1]) < 1e-2)
            msg = f"two-destination={two} flip={flip}"
        elif kind == "f32_split":
            # fp32 tensors, split-f16 products (csrc/common.h SplitF16, the default) against the exact fp32 kernels: every mode of the tile
            # kernel and every weight-gradient mode, channel counts on and off the 16-channel fast-path contract, one or two sources, operand
            # magnitudes drawn over eleven decades (the per-tensor power-of-two scaling must keep fp16's range out of the picture)
            F32 = mia_hip.F32
            ops.F32_SPLIT_MIN_MACS = 0
            sx, sd, sw = (10.0 ** rng.uniform(-7, 4) for _ in range(3))
            mode = rng.choice(["s1", "s1_dgrad", "s2", "s2_dgrad", "t2", "t2_dgrad", "wg_s1", "wg_s2", "wg_t2"])
            cin = rng.choice([16, 32, 48, 64, 96, 128, 160, 256])
            cout = rng.choice([16, 32, 48, 64, 96, 128, 192])
            two = mode in ("s1", "wg_s1") and rng.random() < 0.4
            f = lambda *shape: torch.randn(*shape, device=dev) * sx  # noqa: E731
            hh, ww = (2 * h - rng.randint(0, 1), 2 * w - rng.randint(0, 1)) if mode in ("s2", "s2_dgrad", "wg_s2") else (h, w)
            ho, wo = ((hh + 1) // 2, (ww + 1) // 2) if mode in ("s2", "s2_dgrad", "wg_s2") else (hh, ww)
            wt = torch.randn(cout, cin * (2 if two else 1), 3, 3, device=dev) / (3 * (cin * (2 if two else 1)) ** 0.5) * sw
            wtt = torch.randn(cin, cout, 2, 2, device=dev) / (2 * cin ** 0.5) * sw  # ConvTranspose2d(cin -> cout)
            b = torch.randn(cout, device=dev) * (sx * sw)
            x1, x2 = f(n, hh, ww, cin), (f(n, hh, ww, cin) if two else None)
            dy = torch.randn(n, ho, wo, cout, device=dev) * sd
            if two:
                x2 = x2 * 10.0 ** rng.uniform(-3, 3)
            if mode == "s1":
                fn = lambda: ops.conv_mma(CONV_G3S1, x1, x2, *ops.PackCache().get(wt, F32, True), False, b, cout, (hh, ww), want_stats=True)[0]  # noqa: E731
            elif mode == "s1_dgrad":
                fn = lambda: ops.conv_mma(CONV_G3S1, dy, None, *ops.PackCache().get(wt, F32, False), True, None, cin, (hh, ww))[0]  # noqa: E731
            elif mode == "s2":
                fn = lambda: ops.conv_mma(CONV_G3S2, x1, None, *ops.PackCache().get(wt, F32, True), False, b, cout, (ho, wo), want_stats=True)[0]  # noqa: E731
            elif mode == "s2_dgrad":
                from mia_hip import CONV_T3S2
                fn = lambda: ops.conv_mma(CONV_T3S2, dy, None, *ops.PackCache().get(wt, F32, False), False, None, cin, (hh, ww))[0]  # noqa: E731
            elif mode == "t2":
                fn = lambda: ops.conv_mma(CONV_T2S2, x1, None, *ops.PackCache().get(wtt, F32, False), False, b, cout, (2 * hh, 2 * ww))[0]  # noqa: E731
            elif mode == "t2_dgrad":
                fine = f(n, 2 * hh, 2 * ww, cout)
                fn = lambda: ops.conv_mma(CONV_G2S2, fine, None, *ops.PackCache().get(wtt, F32, True), False, None, cin, (hh, ww))[0]  # noqa: E731
            elif mode == "wg_s1":
                fn = lambda: ops.conv_wgrad(WGRAD_3S1, x1, x2, dy, tuple(wt.shape), cout, cin * (2 if two else 1))  # noqa: E731
            elif mode == "wg_s2":
                fn = lambda: ops.conv_wgrad(WGRAD_3S2, x1, None, dy, tuple(wt.shape), cout, cin)  # noqa: E731
            else:
                fine = f(n, 2 * hh, 2 * ww, cout)
                fn = lambda: ops.conv_wgrad(WGRAD_2S2, fine, None, x1, tuple(wtt.shape), cin, cout)  # noqa: E731
            sv = rng.choice([1, 2])  # four products on interleaved words / three on planes where the conv has 32 x 32 tiles
            r1, r0 = ab("f32_split", sv, fn)
            e = relerr(r1, r0)
            ok = e < 4e-6 and bool(torch.isfinite(r1).all())  # two fp32-accurate evaluations of one sum
            msg = f"f32_split={sv} {mode} cin={cin}{'x2' if two else ''} cout={cout} in {hh}x{ww} scales {sx:.0e}/{sw:.0e}/{sd:.0e} relerr {e:.1e}{' (identical: exact kernel ran)' if e == 0 else ''}"
        elif kind == "wgrad_t2":
            cin, cout = rng.choice([(128, 64), (256, 128), (384, 192), (512, 256)])
            x, dout = t(n, h, w, cin), t(n, 2 * h, 2 * w, cout)
            g1, g0 = ab("wgrad_t2", 1, lambda: ops.conv_wgrad(WGRAD_2S2, dout, None, x, (cin, cout, 2, 2), cin, cout))
            ok = relerr(g1, g0) < 1e-4
            msg = f"cin={cin} cout={cout} relerr {relerr(g1, g0):.1e}"
        else:
            s2 = kind == "wgrad_bt_s2"
            cin, cout = rng.choice([(64, 128), (128, 128), (128, 256), (192, 384)])
            hh, ww = (2 * h - rng.randint(0, 1), 2 * w - rng.randint(0, 1)) if s2 else (h, w)
            ho, wo = ((hh + 1) // 2, (ww + 1) // 2) if s2 else (hh, ww)
            x, dy = t(n, hh, ww, cin), t(n, ho, wo, cout)
            g1, g0 = ab("wgrad_bt", 1, lambda: ops.conv_wgrad(WGRAD_3S2 if s2 else WGRAD_3S1, x, None, dy, (cout, cin, 3, 3), cout, cin))
            ok = relerr(g1, g0) < 1e-4
            msg = f"cin={cin} cout={cout} x {hh}x{ww} relerr {relerr(g1, g0):.1e}"
        bad += 0 if ok else 1
        print(f"[{case:3d}] {kind:13s} n={n} {h}x{w} {msg}: {'ok' if ok else 'MISMATCH'}", flush=True)
    print(f"{a.cases - bad} / {a.cases} cases agree")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
