"""Sustained MFMA rate under the power limit vs MFMA shape and LDS / HBM traffic beside it (tools/probe/mfma_power.hip).
    python tools/probe/mfma_power.py > gpurun_out/mfma_power.txt      (on the GPU box; builds tools/ab/mfma_power.so with hipcc)"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = os.path.join(ROOT, "tools", "ab", "mfma_power.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
src = os.path.join(ROOT, "tools", "probe", "mfma_power.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", src, "-o", so])
lib = ctypes.CDLL(so)
lib.mfma_power_run.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int,
                               ctypes.c_void_p]
dev = torch.device("cuda:0")
units = 1 << 26  # 1 GiB of 16-byte units
buf = torch.randint(0, 2 ** 31 - 1, (units * 4,), dtype=torch.int32, device=dev)
MASK = int(os.environ.get("MASK", "0xBFFFBFFF"), 16)
buf &= MASK - (1 << 32) if MASK >= 1 << 31 else MASK  # default: random sign / mantissa, |x| < 2 (realistic toggle rate); 0x3F803F80: near-constant operands
out = torch.zeros(256, device=dev)
clocks = torch.zeros(128, dtype=torch.int64, device=dev)
blocks = 512
st = torch.cuda.current_stream().cuda_stream
print(f"operand mask {MASK:#x}")
print("shape ldsr/8mfma hbm   ms    TFLOP/s  shader-clock GHz   LDS TB/s   HBM TB/s")
VALU_ONLY = os.environ.get("VALU_ROWS") == "1"  # just the rows with packed-fp32 VALU work beside the canonical block's mix
for hbm, shape in (((2, 0),) if VALU_ONLY else ((0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (4, 0), (8, 0))):
    for valu in ((0, 8, 16, 24) if VALU_ONLY else (0,)):
        for ldsr in ((4,) if VALU_ONLY else (0, 2, 4, 6, 8)):
            iters = 400000
            flop_per_iter = 8 * 16 * 16 * 32 * 2 if shape == 0 else 4 * 32 * 32 * 16 * 2  # per wave
            for rep in range(2):  # first = warm-up into the power-limited state
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = lib.mfma_power_run(shape, ldsr, hbm, valu, buf.data_ptr(), out.data_ptr(), iters, units, clocks.data_ptr(), blocks, st)
                assert rc == 0, rc
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
            tf = flop_per_iter * iters * blocks * 4 / (ms * 1e-3) / 1e12
            c = clocks.cpu()
            ghz = float((c[0::2].double() / (c[1::2].double() / 100e6)).mean()) / 1e9  # wall clock = 100 MHz
            lds = ldsr * 1024 * iters * blocks * 4 / (ms * 1e-3) / 1e12
            hb = (16 * 256 * iters * blocks / hbm / (ms * 1e-3) / 1e12) if hbm else 0.0
            print(f"{'16x16x32' if shape == 0 else '32x32x16'}{'+%dpk' % valu if valu else ''} {ldsr:5d} {hbm:5d} {ms:8.2f} {tf:8.1f} {ghz:10.3f} {lds:14.2f} {hb:10.2f}", flush=True)
