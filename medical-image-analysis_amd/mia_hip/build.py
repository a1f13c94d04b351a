"""Build libmia_hip.so (gfx950) in-tree with hipcc.  `python -m mia_hip.build` or `build()`."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
LIB = os.path.join(HERE, "libmia_hip.so")
OBJ = os.path.join(HERE, "_obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc"]
# Kernels that issue LDS-DMA in inline asm and count `vmcnt` by hand: a register spill would add compiler-made scratch
# traffic to the same counter and break the count, so the build fails unless their scratch size is 0.
COUNTED_VMCNT = {"conv_bt.hip": ("conv_bt_kernel",), "conv_pw.hip": ("conv_pw_kernel",), "conv64_dma.hip": ("conv64_dma_kernel",), "conv_wgrad.hip": ("wgrad_bf16_dma_kernel", "wgrad_bf16_dma96_kernel", "wgrad_bf16_bt_kernel", "wgrad_bf16_bt_s2_kernel", "wgrad_bf16_bt_t2_kernel")}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def check_no_scratch(src, remarks, kernels, obj):
    """Parse hipcc's kernel-resource-usage remarks: every kernel whose mangled name contains one of `kernels` must report
    `ScratchSize [bytes/lane]: 0` (and must be present at all)."""
    name, seen = None, set()
    for line in remarks.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split()[0]
        elif "ScratchSize" in line and name and any(k in name for k in kernels):
            seen.add(name)
            size = int(line.split("ScratchSize [bytes/lane]:")[1].split()[0])
            if size != 0:
                if os.path.exists(obj):
                    os.remove(obj)
                raise RuntimeError(f"{os.path.basename(src)}: {name} spills {size} bytes/lane to scratch; its hand-counted "
                                   f"vmcnt pipeline would be wrong")
    if not seen:
        raise RuntimeError(f"{os.path.basename(src)}: no resource-usage remark found for {kernels}")


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    srcs = sources()
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        guarded = COUNTED_VMCNT.get(os.path.basename(s), ())
        cmd = [hipcc] + FLAGS + (["-Rpass-analysis=kernel-resource-usage"] if guarded else []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        if guarded:
            check_no_scratch(s, r.stderr, guarded, o)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
