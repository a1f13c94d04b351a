#!/usr/bin/env python
"""Scan the gfx950 ISA of every csrc/*.hip for the store-data hazard found in round 4 (csrc/conv64.hip, CR epilogue):
a `buffer_store_dwordx3/x4` / `global_store_dwordx3/x4` / `scratch_store` whose DATA (or address) registers are written again by
DATA registers are written again fewer than `WINDOW` wait states later.  hipcc pads this hazard to 2 wait states; on MI355X a VALU
write 2 wait states behind a 16-byte store still reached the store's last lane phase in ~6 % of the tiles (lanes 12-15 of every
16; tools/probe/diag_cr3.py, diag_cr4.py), 4 wait states were always enough.  The number printed is the wait-state distance.

    python tools/check_store_hazard.py [--window 4] [file.hip ...]      exit code 1 if anything is found
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "medical-image-analysis_amd", "csrc")
REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def dest_regs(line):
    """VGPRs an instruction writes (first operand of v_* / loads; BOTH operands of v_permlane*_swap / v_swap)."""
    parts = line.split(None, 1)
    if len(parts) < 2:
        return set()
    op, rest = parts
    ops = [o.strip() for o in rest.split(",")]
    if op.startswith(("buffer_store", "global_store", "scratch_store", "ds_write", "flat_store", "s_", "v_cmp", "v_nop", "buffer_wbl2", "buffer_inv")):
        if op.startswith("v_cmp") and ops and ops[0].startswith("v"):
            return regs(ops[0])
        return set()
    if op.startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap")):
        return regs(ops[0]) | regs(ops[1])
    if op.startswith(("v_", "buffer_load", "global_load", "scratch_load", "ds_read", "flat_load")):
        if "lds" in rest.split() and op.startswith("buffer_load"):
            return set()
        return regs(ops[0])
    return set()


def scan(path, window):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-S", "--cuda-device-only", "-o", out, path],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
        lines = open(out).read().splitlines()
    found, kernel = [], "?"
    body = []
    for ln in lines:
        s = ln.strip()
        if s.endswith(":") and s.startswith("_Z") and "@" not in s:
            kernel = s[:-1]
        if ln.startswith("_Z") and ":" in ln:
            kernel = ln.split(":")[0]
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        body.append((kernel, s.split(";")[0].strip()))
    for i, (k, ins) in enumerate(body):
        op = ins.split(None, 1)[0]
        if not (op.startswith(("buffer_store_dwordx3", "buffer_store_dwordx4", "global_store_dwordx3", "global_store_dwordx4"))):
            continue
        ops = [o.strip() for o in ins.split(None, 1)[1].split(",")]
        data = regs(ops[0]) if op.startswith("buffer_store") else regs(ops[1])
        addr = regs(ops[1]) if op.startswith("buffer_store") else regs(ops[0])
        ws = 0  # wait states between the store and instruction i + j (s_nop N counts N + 1, everything else 1)
        for j in range(1, 3 * window):
            if i + j >= len(body) or body[i + j][0] != k or ws >= window:
                break
            nxt = body[i + j][1]
            if nxt.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_barrier", "s_waitcnt")):
                break  # (a wait / barrier / branch: unknown but long)
            w = dest_regs(nxt)
            if w & data:
                found.append((k, ins, ws, nxt, "data"))
                break
            m = re.match(r"s_nop\s+(\d+)", nxt)
            ws += int(m.group(1)) + 1 if m else 1
    return found


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=int, default=4)
    ap.add_argument("files", nargs="*")
    a = ap.parse_args()
    files = a.files or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    for f in files:
        hits = scan(f, a.window)
        print(f"{os.path.basename(f)}: {len(hits)} 12/16-byte store(s) whose data registers are rewritten fewer than {a.window} wait states later")
        for k, ins, j, nxt, what in hits[:40]:
            print(f"   {k[:60]}: `{ins}` -> {j} wait states -> `{nxt}`")
        bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
