#!/usr/bin/env python
"""Summarise the rocprofv3 PMC passes of tools/pmc_conv64.sh into profiles/: one CSV of per-launch counter means for the
canonical conv launch plus the JSON that bench.py reads for `roofline.traffic`.

    python tools/pmc_summarise.py <tag> [kernel-substring]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    needle = sys.argv[2] if len(sys.argv) > 2 else "conv64"
    label = sys.argv[3] if len(sys.argv) > 3 else "canonical_conv"
    out_rows, means = [], {}
    for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_*/"))):
        f = os.path.join(d, "p_counter_collection.csv")
        if not os.path.exists(f):
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        kt = [r for r in csv.DictReader(open(os.path.join(d, "p_kernel_trace.csv"))) if needle in r["Kernel_Name"]]
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in kt]
        for k, v in sorted(agg.items()):
            means[k] = sum(v) / len(v)
            out_rows.append({"pass": os.path.basename(d.rstrip("/")), "kernel": needle, "counter": k, "mean_per_launch": means[k],
                             "launches": len(v), "avg_duration_us": sum(dur) / max(1, len(dur))})
    dst = os.path.join(ROOT, "profiles", f"{tag}_pmc_{label}.csv")
    with open(dst, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(out_rows[0].keys()))
        w.writeheader()
        w.writerows(out_rows)
    print("wrote", dst)
    if "FETCH_SIZE" in means and "WRITE_SIZE" in means and label == "canonical_conv":
        rec = {"config": "cfg3", "batch": 32, "dtype": "bf16", "kernel": needle, "FETCH_SIZE_KB": means["FETCH_SIZE"],
               "WRITE_SIZE_KB": means["WRITE_SIZE"],
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on tools/microbench.py conv --c 64 --size 512 --batch 32; "
                       "HBM bytes = 2 x FETCH_SIZE (gfx950 counts 128-B requests at 64 B) + WRITE_SIZE, KB -> bytes x 1024"}
        j = os.path.join(ROOT, "profiles", "r02_pmc_canonical_conv.json")
        json.dump(rec, open(j, "w"), indent=1)
        print("wrote", j, "->", (2 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024 / 1e9, "GB per launch")


if __name__ == "__main__":
    main()
