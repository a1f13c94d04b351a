#!/usr/bin/env python
"""Summarise tools/r4_pmc_block.sh into profiles/<tag>_pmc_canonical_block.{csv,json}: per kernel of the canonical block
(forward: conv (+ normalise-on-load), norm_finalize, norm_act_fwd; backward: colreduce, norm_bwd_finalize, norm_act_bwd,
weight gradient, its slab reduce, input gradient) the mean duration and the mean of every collected counter, and the HBM
bytes per launch = 2 x FETCH_SIZE (gfx950 counts a 128-byte request as 64) + WRITE_SIZE, KB x 1024.  bench.py reads the
JSON for `roofline.traffic`.      python tools/r4_pmc_block.py <tag> [out-tag]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"conv64_persist_kernel<true>": "conv_nl", "conv64_persist_kernel<false>": "conv_or_dgrad", "norm_fwd_sum_kernel": "norm_finalize",
           "norm_finalize_kernel": "norm_finalize2", "norm_act_fwd_stream_kernel": "norm_act_fwd", "colreduce_vec_kernel": "bwd_colreduce",
           "norm_bwd_finalize_kernel": "bwd_finalize", "norm_bwd_sum_kernel": "bwd_sum", "norm_act_bwd_stream_kernel": "norm_act_bwd",
           "wgrad_bf16_2wg_kernel<8, true>": "wgrad_nl", "wgrad_bf16_dma_kernel": "wgrad_dma", "wgrad_reduce_small4_kernel": "wgrad_reduce"}


def short(name):
    for k, v in KERNELS.items():
        if k in name:
            return v
    return None


def main():
    tag = sys.argv[1]
    out_tag = sys.argv[2] if len(sys.argv) > 2 else tag
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_*/"))):
        f = os.path.join(d, "p_counter_collection.csv")
        if not os.path.exists(f):
            continue
        seen = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for r in csv.DictReader(open(os.path.join(d, "p_kernel_trace.csv"))):
            k = short(r["Kernel_Name"])
            if k is not None:
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows, rec = [], {"config": "cfg3", "batch": 32, "dtype": "bf16", "shape": "64 -> 64 @ 512 x 512 x 32", "kernels": {},
                     "note": "rocprofv3 --kernel-trace --pmc <one group per pass> on tools/microbench.py block --c 64 --size 512 --batch 32; "
                             "hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE counts 128-byte requests at 64 B)"}
    for k in sorted(per):
        means = {c: sum(v) / len(v) for c, v in per[k].items()}
        d_us = sum(dur[k]) / max(1, len(dur[k]))
        for c, v in sorted(means.items()):
            rows.append({"kernel": k, "counter": c, "mean_per_launch": v, "launches_sampled": len(per[k][c]), "avg_duration_us_under_pmc": round(d_us, 1)})
        e = {"avg_duration_us_under_pmc": round(d_us, 1)}
        if "FETCH_SIZE" in means and "WRITE_SIZE" in means:
            e.update(FETCH_SIZE_KB=means["FETCH_SIZE"], WRITE_SIZE_KB=means["WRITE_SIZE"],
                     hbm_bytes=(2 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024)
        if "SQ_INSTS_MFMA" in means and means["SQ_INSTS_MFMA"] > 0:
            e["valu_per_mfma"] = round(means.get("SQ_INSTS_VALU", 0) / means["SQ_INSTS_MFMA"], 3)
            e["lds_per_mfma"] = round(means.get("SQ_INSTS_LDS", 0) / means["SQ_INSTS_MFMA"], 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in means and means.get("SQ_BUSY_CYCLES"):
            # SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader engines: / 32 = per-SIMD busy share
            # (the same normalisation reproduces the 0.60 of profiles/r02b_pmc_canonical_conv.csv for the plain kernel)
            e["mfma_busy_frac"] = round(means["SQ_VALU_MFMA_BUSY_CYCLES"] / means["SQ_BUSY_CYCLES"] / 32.0, 4)
        rec["kernels"][k] = e
    dst = os.path.join(ROOT, "profiles", f"{out_tag}_pmc_canonical_block.csv")
    with open(dst, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    json.dump(rec, open(os.path.join(ROOT, "profiles", f"{out_tag}_pmc_canonical_block.json"), "w"), indent=1)
    for k, e in rec["kernels"].items():
        print(f"{k:16s} {e.get('avg_duration_us_under_pmc', 0):9.1f} us  hbm {e.get('hbm_bytes', 0) / 1e9:6.3f} GB  {e}")


if __name__ == "__main__":
    main()
