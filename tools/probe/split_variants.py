"""Timing-only variants of the split-mode tile conv (wrong results on purpose) to see what bounds it: builds tools/ab/libmia_sv<k>.so from
patched copies of csrc/conv_mma_fast.hip.   python tools/probe/split_variants.py   (dev container; then run tools/probe/split_variants.sh on the GPU)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "medical-image-analysis_amd", "csrc")
OBJ = os.path.join(ROOT, "medical-image-analysis_amd", "mia_hip", "_obj")
AB = os.path.join(ROOT, "tools", "ab")
src = open(os.path.join(CSRC, "conv_mma_fast.hip")).read()
A_SPLIT = "= SPLIT ? SplitBf16::unit(pa[i]) : pa[i];"
B_SPLIT = "= SPLIT ? SplitBf16::unit(pb[i]) : pb[i];"
DUP = "const u32x4 bh = SplitBf16::dup_hi(bfr[n]), bl = SplitBf16::dup_lo(bfr[n]);"
assert all(s in src for s in (A_SPLIT, B_SPLIT, DUP))
variants = {
    1: [(B_SPLIT, "= pb[i];")],
    2: [(DUP, "const u32x4 bh = bfr[n], bl = bfr[n];")],
    3: [(B_SPLIT, "= pb[i];"), (DUP, "const u32x4 bh = bfr[n], bl = bfr[n];")],
    4: [(B_SPLIT, "= pb[i];"), (DUP, "const u32x4 bh = bfr[n], bl = bfr[n];"), (A_SPLIT, "= pa[i];")],
}
os.makedirs(AB, exist_ok=True)
objs = [os.path.join(OBJ, f) for f in os.listdir(OBJ) if f.endswith(".o") and f != "conv_mma_fast.o"]
for k, reps in variants.items():
    s = src
    for a, b in reps:
        s = s.replace(a, b)
    path = os.path.join(CSRC, f"_sv{k}.hip")  # beside the headers it includes
    open(path, "w").write(s)
    try:
        o = os.path.join(AB, f"sv{k}.o")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-c", path, "-o", o])
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(AB, f"libmia_sv{k}.so"), o] + objs)
    finally:
        os.remove(path)
    print("built variant", k, flush=True)
