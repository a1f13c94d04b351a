"""ctypes binding of libmia_hip.so (include/mia_hip.h).

The prototypes are parsed from the header itself, so the header is the single source of truth for
the C ABI.  There is NO CPU fallback: if the shared object is missing or a call fails, the caller
gets an exception.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HEADER = os.path.join(ROOT, "include", "mia_hip.h")
LIB_PATH = os.environ.get("MIA_HIP_LIB") or os.path.join(HERE, "libmia_hip.so")  # MIA_HIP_LIB: A/B a differently built library

F32, BF16 = 0, 1
CONV_G3S1, CONV_G3S2, CONV_G2S2, CONV_T3S2, CONV_T2S2, CONV_G1 = range(6)
WGRAD_3S1, WGRAD_3S2, WGRAD_2S2 = range(3)
NORM_INSTANCE, NORM_BATCH = 0, 1
LOSS_SOFTMAX, LOSS_DO_BG, LOSS_BATCH, LOSS_SQUARED, LOSS_DENSE = 1, 2, 4, 8, 16
OPT_ADAM, OPT_ADAMW, OPT_SGD = 0, 1, 2


class MiaError(RuntimeError):
    pass


_CTYPES = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64, "float": ctypes.c_float, "double": ctypes.c_double,
}


def parse_defines(path: str = HEADER) -> Dict[str, int]:
    """{name: value} for every integer `#define MIA_*` in include/mia_hip.h (the constants above must agree with it)."""
    out = {}
    with open(path) as fh:
        src = fh.read()
    for m in re.finditer(r"^#define\s+(MIA_\w+)\s+(-?\d+)\b", src, flags=re.M):
        out[m.group(1)] = int(m.group(2))
    return out


def parse_header(path: str = HEADER) -> Dict[str, Tuple[object, List[object]]]:
    """{name: (restype, [argtypes])} for every prototype in include/mia_hip.h."""
    with open(path) as fh:
        txt = fh.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = "\n".join(l for l in txt.splitlines() if not l.strip().startswith("#"))
    protos = {}
    for m in re.finditer(r"(const char\*|int)\s+(mia_\w+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPES[a.split()[-2] if len(a.split()) > 1 else a])
        protos[name] = (ctypes.c_char_p if ret.startswith("const char") else ctypes.c_int, argtypes)
    return protos


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MiaError(f"{LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME as
        # /opt/rocm's).  Load torch's copy first so the tensors' allocator/streams and our kernels
        # share a runtime; loading ours first would bind everyone to the system copy.
        import torch  # noqa: F401
        tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(tl):
            ctypes.CDLL(tl, mode=ctypes.RTLD_GLOBAL)
        l = ctypes.CDLL(LIB_PATH)
        for name, (ret, args) in parse_header().items():
            fn = getattr(l, name)
            fn.restype = ret
            fn.argtypes = args
        exp = os.path.join(os.path.dirname(HEADER), "mia_hip_experiments.h")  # probe builds (-DMIA_EXPERIMENTS) export a few more
        if os.path.exists(exp):
            for name, (ret, args) in parse_header(exp).items():
                if hasattr(l, name):
                    fn = getattr(l, name)
                    fn.restype = ret
                    fn.argtypes = args
        # A/B knobs (kernel selection only, never numerics contracts): MIA_OPTIONS="wgrad_dma=0,conv64=1" -> mia_set_option
        for item in filter(None, os.environ.get("MIA_OPTIONS", "").split(",")):
            key, _, val = item.partition("=")
            if l.mia_set_option(key.strip().encode(), int(val or 1)) != 0:
                raise MiaError(f"MIA_OPTIONS: unknown option {key!r}")
        _lib = l
    return _lib


def set_option(name: str, value: int) -> None:
    """Library option (include/mia_hip.h: mia_set_option) -- kernel selection / launch shape for A/B runs, thread-safe."""
    check(lib().mia_set_option(name.encode(), int(value)), "mia_set_option")


def get_option(name: str) -> int:
    v = ctypes.c_int(0)
    check(lib().mia_get_option(name.encode(), ctypes.byref(v)), "mia_get_option")
    return v.value


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().mia_last_error()
        raise MiaError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def call(name: str, *args) -> None:
    check(getattr(lib(), name)(*args), name)
