#!/usr/bin/env python
"""Diagnostic: where a tile of the persistent 64-channel conv spends its cycles (in-kernel s_memtime stamps).

Builds a SEPARATE library with -DCONV64_STAMPS into tools/ab/ (never the shipped one), runs the canonical launch and
prints the per-wave mean cycles per tile of each phase.  Read the SHARES, not the length (the stamps serialise)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "medical-image-analysis_amd")
OUT = os.path.join(ROOT, "tools", "ab")


def build():
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, "libmia_hip_stamps.so")
    srcs = sorted(os.path.join(PKG, "csrc", f) for f in os.listdir(os.path.join(PKG, "csrc")) if f.endswith(".hip"))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-shared", "-DCONV64_STAMPS",
           "-o", lib] + srcs
    subprocess.run(cmd, check=True)
    return lib


def wgrad():
    os.environ["MIA_HIP_LIB"] = os.path.join(OUT, "libmia_hip_stamps.so")
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    import torch
    import mia_hip
    from mia_hip import WGRAD_3S1, ops
    dev = torch.device("cuda:0")
    x = torch.randn(32, 512, 512, 64, device=dev).to(torch.bfloat16)
    dy = torch.randn(32, 512, 512, 64, device=dev).to(torch.bfloat16)
    for _ in range(3):
        ops.conv_wgrad(WGRAD_3S1, x, None, dy, (64, 64, 3, 3), 64, 64)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv_wgrad(WGRAD_3S1, x, None, dy, (64, 64, 3, 3), 64, 64)
    e1.record()
    torch.cuda.synchronize()
    print(f"one call (kernel + slab reduce) with stamps: {e0.elapsed_time(e1):.3f} ms")
    l = ctypes.CDLL(os.environ["MIA_HIP_LIB"])
    print("resident workgroups per CU (runtime occupancy query): dma", l.mia_wgrad_debug_occupancy(0), "bt", l.mia_wgrad_debug_occupancy(1),
          "bt_s2", l.mia_wgrad_debug_occupancy(2), "2wg", l.mia_wgrad_debug_occupancy(3))
    buf = np.zeros(512 * 4 * 8, dtype=np.uint64)
    assert l.mia_wgrad_debug_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    d = buf.reshape(512, 4, 8).astype(np.float64)
    n = d[..., 5]
    live = n > 0
    names = ["barrier 1 (readers done)", "commit (ds_write)", "barrier 2", "fetch issue", "MFMA loop"]
    if os.environ.get("MIA_STAMPS_DMA", "1") != "0":  # wgrad_bf16_dma_kernel's phases
        names = ["DMA issue (tile t+2)", "fragment reads + MFMAs", "vmcnt wait (tile t+1)", "barrier", "-"]
    tot = 0.0
    for i, nm in enumerate(names):
        per = (d[..., i][live] / n[live]).mean()
        tot += per
        print(f"{nm:26s} {per:9.0f} cycles / tile / wave")
    print(f"{'sum':26s} {tot:9.0f}   (tiles per workgroup {n[live].mean():.1f}; MFMA issue floor 144 x 16 = 2304)")
    if d[..., 7][live].min() > 0:
        clk = (d[..., 6][live] / d[..., 7][live]) * 0.1
        print(f"loop of one workgroup: {(d[..., 7][live] * 0.01).mean():.1f} us wall (s_memrealtime), in-kernel clock {np.median(clk):.2f} GHz")


def bt():
    """Phases of the big-tile kernel (conv_bt.hip): python tools/conv64_stamps.py bt [channels]"""
    os.environ["MIA_HIP_LIB"] = os.path.join(OUT, "libmia_hip_stamps.so")
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    import torch
    import mia_hip
    from mia_hip import CONV_G3S1, ops
    dev = torch.device("cuda:0")
    cs = [int(v) for v in sys.argv[sys.argv.index("bt") + 1:]] or [128, 256, 1024]
    for c in cs:
        s = 32768 // c
        x = torch.randn(32, s, s, c, device=dev).to(torch.bfloat16)
        w = (torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5))
        b = torch.randn(c, device=dev)
        wp, npad, kpad = ops.PackCache().get(w, mia_hip.BF16, True)
        for _ in range(3):
            ops.conv_mma(CONV_G3S1, x, None, wp, npad, kpad, False, b, c, (s, s), want_stats=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            ops.conv_mma(CONV_G3S1, x, None, wp, npad, kpad, False, b, c, (s, s), want_stats=True)
        e1.record()
        torch.cuda.synchronize()
        wall_ms = e0.elapsed_time(e1) / 4
        l = ctypes.CDLL(os.environ["MIA_HIP_LIB"])
        buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
        assert l.mia_conv_bt_debug_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        d = buf.reshape(256, 8, 8).astype(np.float64)
        steps, tiles = d[..., 6], d[..., 7]
        live = steps > 0
        print(f"--- conv_bt c={c} {s}x{s} x32: steps/wg {steps[live].mean():.0f}, tiles/wg {tiles[live].mean():.1f}  (MFMA floor per step and wave: 96 x 16 = 1536 cycles)")
        nfin = 3.0 * tiles  # steps that are a tile's final step: 1 per tile (d[6] counts all steps)
        print(f"{'final step (MFMAs + stores)':26s} {(d[..., 0][live] / tiles[live]).mean():9.0f} cycles / tile / wave   (waves 0-3 {(d[:, :4, 0][live[:, :4]] / tiles[:, :4][live[:, :4]]).mean():.0f}, 4-7 {(d[:, 4:, 0][live[:, 4:]] / tiles[:, 4:][live[:, 4:]]).mean():.0f})")
        for i, nm in enumerate(["-", "fragment reads + MFMAs", "vmcnt wait", "barrier"]):
            if i == 0:
                continue
            den = steps - (tiles if i == 1 else 0)
            print(f"{nm:26s} {(d[..., i][live] / den[live]).mean():9.0f} cycles / step / wave   (waves 0-3 {(d[:, :4, i][live[:, :4]] / den[:, :4][live[:, :4]]).mean():.0f}, 4-7 {(d[:, 4:, i][live[:, 4:]] / den[:, 4:][live[:, 4:]]).mean():.0f})")
        for i, nm in ((4, "statistics epilogue"), (5, "next-tile prep")):
            print(f"{nm:26s} {(d[..., i][live] / tiles[live]).mean():9.0f} cycles / tile / wave")
        cyc = d[..., :6].sum(-1)[live].mean()  # the stamped phases cover the loop of a workgroup
        print(f"launch {wall_ms:.3f} ms (stamped build), {cyc:.0f} stamped cycles per wave -> in-kernel clock ~ {cyc / wall_ms / 1e6:.2f} GHz; "
              f"matrix pipe busy {2 * 96 * 16 * steps[live].mean() / cyc * 100:.0f} % of the cycles (two waves per SIMD)")


def c64dma():
    """Phases of conv64_dma_kernel: python tools/conv64_stamps.py c64dma [dgrad]"""
    os.environ["MIA_HIP_LIB"] = os.path.join(OUT, "libmia_hip_stamps.so")
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    import torch
    import mia_hip
    from mia_hip import CONV_G3S1, ops
    dev = torch.device("cuda:0")
    x = torch.randn(32, 512, 512, 64, device=dev).to(torch.bfloat16)
    w = (torch.randn(64, 64, 3, 3, device=dev) / 24)
    b = torch.randn(64, device=dev)
    dg = "dgrad" in sys.argv
    mia_hip.set_option("conv64_dma", 2)
    wp, npad, kpad = ops.PackCache().get(w, mia_hip.BF16, not dg)
    for _ in range(3):
        ops.conv_mma(CONV_G3S1, x, None, wp, npad, kpad, dg, None if dg else b, 64, (512, 512), want_stats=not dg)
    torch.cuda.synchronize()
    l = ctypes.CDLL(os.environ["MIA_HIP_LIB"])
    buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
    assert l.mia_conv64_dma_debug_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    d = buf.reshape(256, 8, 8).astype(np.float64)
    n = d[..., 5]
    live = n > 0
    print(f"--- conv64_dma {'dgrad' if dg else 'fwd+stats'}: tiles/wg {n[live].mean():.0f} (MFMA floor per tile and wave: 144 x 16 = 2304 cycles)")
    for i, nm in enumerate(["head (combine, decode)", "reads + MFMAs (+ slices)", "tail (copy / epilogue)", "vmcnt + lgkm wait", "barrier"]):
        print(f"{nm:26s} {(d[..., i][live] / n[live]).mean():9.0f} cycles / tile / wave   (waves 0-3 {(d[:, :4, i][live[:, :4]] / n[:, :4][live[:, :4]]).mean():.0f}, 4-7 {(d[:, 4:, i][live[:, 4:]] / n[:, 4:][live[:, 4:]]).mean():.0f})")


def main():
    if "c64dma" in sys.argv:
        return c64dma()
    if "bt" in sys.argv:
        return bt()
    if "build" in sys.argv:
        print(build())
        return
    if "wgrad" in sys.argv:
        return wgrad()
    os.environ["MIA_HIP_LIB"] = os.path.join(OUT, "libmia_hip_stamps.so")
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    import torch
    import mia_hip
    from mia_hip import CONV_G3S1, ops
    dev = torch.device("cuda:0")
    x = torch.randn(32, 512, 512, 64, device=dev).to(torch.bfloat16)
    w = (torch.randn(64, 64, 3, 3, device=dev) / 24)
    b = torch.randn(64, device=dev)
    wp, npad, kpad = ops.PackCache().get(w, mia_hip.BF16, True)
    for _ in range(3):
        ops.conv_mma(CONV_G3S1, x, None, wp, npad, kpad, False, b, 64, (512, 512), want_stats=True)
    torch.cuda.synchronize()
    l = ctypes.CDLL(os.environ["MIA_HIP_LIB"])
    print("resident workgroups per CU (runtime occupancy query): dma", l.mia_wgrad_debug_occupancy(0), "bt", l.mia_wgrad_debug_occupancy(1),
          "bt_s2", l.mia_wgrad_debug_occupancy(2), "2wg", l.mia_wgrad_debug_occupancy(3))
    buf = np.zeros(512 * 4 * 8, dtype=np.uint64)
    rc = l.mia_conv64_debug_read(buf.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    d = buf.reshape(512, 4, 8).astype(np.float64)
    n = d[..., 5]
    names = ["fetch issue", "MFMA loop", "epilogue", "barrier 1 (wait)", "commit + barrier 2"]
    tot = 0.0
    for i, nm in enumerate(names):
        per = (d[..., i] / np.maximum(n, 1)).mean()
        tot += per
        print(f"{nm:20s} {per:9.0f} cycles / tile / wave")
    print(f"{'sum':20s} {tot:9.0f}   (tiles per workgroup: {n.mean():.1f}; MFMA issue floor 288 x 16 = 4608)")


if __name__ == "__main__":
    main()
