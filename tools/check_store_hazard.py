#!/usr/bin/env python
"""Scan the gfx950 ISA of the BUILT library for the store-data hazard found in round 4 (csrc/conv64.hip, column-reduce epilogue):
a 12- / 16-byte vector store whose DATA registers are overwritten by a VALU instruction fewer than WINDOW wait states later.
hipcc's own rule leaves 1-2 wait states; on MI355X a VALU write 2 wait states behind a 16-byte store, beside a dozen in-flight
loads of the same wave, reached the store's last lane phase in ~7 % of the tiles (lanes 12-15 of every 16 stored the NEW value;
profiles/r05_store_hazard.txt: 22 794 bad tiles of 327 680 unpadded, 0 with 2 or more extra wait states).

Findings come in two classes:
  valu   the overwriting instruction is a VALU / permlane / readlane op: it writes the register a fixed few cycles after issue.
         These are the hazard.  Listed with their wait-state distance and counted per (kernel, distance).
  async  the overwriting instruction is a load (ds_read / buffer_load / global_load / scratch_load) or an MFMA: its result arrives
         tens to hundreds of cycles later.  Reported with --async for completeness, never counted.

Distances count instructions: the overwriting instruction is the d-th after the store (s_nop N counts N + 1).  hipcc's own
rule for gfx950 (GCNHazardRecognizer: 2 wait states between a > 64-bit store and a VALU write of its data) puts compiler-made code at
d >= 3; the site that failed on hardware sat exactly there, with twelve loads of the same wave in flight.

    python tools/check_store_hazard.py [--window 4] [--lib path.so] [--async] [--write-allow]
Exit code 1 when (a) a finding sits at d < 3 -- closer than the compiler itself would put it: hand-written asm next to a store -- or
(b) a kernel FAMILY (template name) that tools/store_hazard_allow.json does not list has a finding at d = 3.  The allow-list names the
families whose d = 3 sites are compiler-made and whose outputs a bit-exact full-size GPU test pins (so a wrong lane phase would show);
tests/test_abi.py runs this check on the built library, so a NEW family with such a site fails the CPU suite and gets looked at
(pad it like csrc/conv64.hip does, or list it with the test that covers it).  --write-allow adds the current families (edit the reasons).
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "medical-image-analysis_amd", "mia_hip", "libmia_hip.so")
ALLOW = os.path.join(ROOT, "tools", "store_hazard_allow.json")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
REG = re.compile(r"v\[(\d+):(\d+)\]|\bv(\d+)\b")
STORES = ("buffer_store_dwordx3", "buffer_store_dwordx4", "global_store_dwordx3", "global_store_dwordx4", "flat_store_dwordx3",
          "flat_store_dwordx4", "scratch_store_dwordx3", "scratch_store_dwordx4")
ASYNC = ("buffer_load", "global_load", "flat_load", "scratch_load", "ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append",
         "buffer_atomic", "global_atomic", "flat_atomic", "ds_add_rtn", "ds_max_rtn", "ds_min_rtn", "image_")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def dest(op, operands):
    """(VGPRs written, class) of one instruction; class 'valu' | 'async' | None."""
    if not operands:
        return set(), None
    if op.startswith(("buffer_store", "global_store", "flat_store", "scratch_store", "ds_write", "s_", "v_nop", "buffer_wbl2", "buffer_inv",
                      "ds_nop", "v_cmpx")):
        return set(), None
    if op.startswith("v_cmp"):
        return (regs(operands[0]), "valu") if operands[0].startswith("v") else (set(), None)
    if op.startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap")):
        return regs(operands[0]) | regs(operands[1]), "valu"
    if op.startswith(ASYNC):
        if op.startswith("buffer_load") and "lds" in operands[-1].split():
            return set(), None
        return regs(operands[0]), "async"
    if op.startswith(("v_mfma", "v_smfmac")):
        return regs(operands[0]), "async"  # the result lands >= 4 passes (16+ cycles) after issue: far beyond a store's lane phases
    if op.startswith("v_"):
        return regs(operands[0]), "valu"
    return set(), None


def disassemble(lib):
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, os.path.basename(lib))
        os.symlink(os.path.abspath(lib), tmp)
        subprocess.run([OBJDUMP, "--offloading", tmp], check=True, capture_output=True)
        cos = [os.path.join(td, f) for f in os.listdir(td) if "gfx950" in f]
        text = ""
        for co in sorted(cos):
            text += subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    return text


def demangle(names):
    if not names:
        return {}
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.splitlines()))


def scan(text, window):
    """-> list of (kernel, class, distance, store line, writer line)."""
    found, kernel, insts = [], None, []

    def flush():
        for i, (op, operands, line) in enumerate(insts):
            if not op.startswith(STORES):
                continue
            data = regs(operands[0]) if op.startswith("buffer_store") else regs(operands[1] if len(operands) > 1 else "")
            if not data:
                continue
            ws = 0
            for op2, operands2, line2 in insts[i + 1:]:
                ws += 1
                if ws >= window:
                    break
                d, cls = dest(op2, operands2)
                if cls and d & data:
                    found.append((kernel, cls, ws, line, line2))
                    break
                if op2 == "s_nop":
                    ws += int(operands2[0], 0) if operands2 else 0
                if op2.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
                    break

    for raw in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", raw)
        if m:
            if kernel is not None:
                flush()
            kernel, insts = m.group(1), []
            continue
        line = raw.split("//")[0].strip()
        if not line or kernel is None:
            continue
        parts = line.split(None, 1)
        op = parts[0]
        operands = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        insts.append((op, operands, line))
    if kernel is not None:
        flush()
    return found


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=int, default=4)
    ap.add_argument("--lib", default=LIB)
    ap.add_argument("--async", dest="show_async", action="store_true")
    ap.add_argument("--write-allow", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args()
    found = scan(disassemble(args.lib), args.window)
    names = demangle(sorted({f[0] for f in found}))
    counts = collections.defaultdict(lambda: collections.Counter())
    for k, cls, ws, st, wr in found:
        if cls == "valu":
            counts[short(names.get(k, k))][str(ws)] += 1
    if not args.quiet:
        for k, cls, ws, st, wr in found:
            if cls == "valu" or args.show_async:
                print(f"{cls:5s} {ws} wait states  {short(names.get(k, k))[:90]}\n        {st}\n        {wr}")
    n_valu = sum(1 for f in found if f[1] == "valu")
    n_async = len(found) - n_valu
    print(f"{n_valu} valu finding(s), {n_async} async (not counted) within {args.window} wait states of a 12/16-byte store")
    fam = lambda k: k.split("<")[0]
    cur = {k: dict(v) for k, v in sorted(counts.items())}
    allow = json.load(open(ALLOW)) if os.path.exists(ALLOW) else {"window": args.window, "families": {}}
    if args.write_allow:
        for k in cur:
            allow["families"].setdefault(fam(k), "compiler-made d = 3 sites; covered by: (name the bit-exact test)")
        allow["window"] = args.window
        json.dump(allow, open(ALLOW, "w"), indent=1, sort_keys=True)
        print("wrote", ALLOW)
        return 0
    bad = []
    for k, per in cur.items():
        for ws, n in per.items():
            if int(ws) < 3:
                bad.append(f"{k}: {n} site(s) at {ws} wait states -- closer than hipcc's own rule (hand-written asm beside a store?)")
            elif fam(k) not in allow["families"]:
                bad.append(f"{k}: {n} site(s) at {ws} wait states in a kernel family the allow-list does not know")
    for b_ in bad:
        print("NEW store-data hazard site:", b_)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
