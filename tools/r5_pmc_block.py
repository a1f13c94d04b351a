#!/usr/bin/env python
"""Summarise tools/r5_pmc_block.sh: per kernel of the canonical block the mean duration and HBM bytes per launch =
(2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md: KB units; gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes).
    python tools/r5_pmc_block.py <cfg> <C> <size> <batch> <dtype>    -> profiles/r05_pmc_canonical_block_<cfg>.{csv,json}"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROLES = (("conv64_persist_kernel<true", "conv_nl"), ("conv64_persist_kernel<false", "conv_or_dgrad"), ("conv_bt_kernel", "conv_or_dgrad"),
         ("conv_mma_fast_kernel", "conv_or_dgrad"), ("norm_fwd_sum_kernel", "norm_finalize"), ("norm_finalize_kernel", "norm_finalize2"),
         ("norm_act_fwd_stream_kernel", "norm_act_fwd"), ("colreduce_vec_kernel", "bwd_colreduce"), ("norm_bwd_finalize_kernel", "bwd_finalize"),
         ("norm_bwd_sum_kernel", "bwd_sum"), ("norm_act_bwd_stream_kernel", "norm_act_bwd"), ("wgrad_bf16_2wg_kernel<8, true>", "wgrad_nl"),
         ("wgrad_bf16_dma96_kernel", "wgrad"), ("wgrad_bf16_dma_kernel", "wgrad"), ("wgrad_bf16_bt_kernel", "wgrad"), ("wgrad_f32_fast_kernel", "wgrad"), ("wgrad_reduce", "wgrad_reduce"),
         ("amax_kernel", "amax"))


def role(name):
    for k, v in ROLES:
        if k in name:
            return v
    return None


def main():
    cfg, c, size, batch, dt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    names = {}
    for pmc in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(ROOT, "gpurun_out", f"pmc5_{cfg}_{pmc}")
        f = None
        for root, _, files in os.walk(d):
            for fn in files:
                if fn.endswith("counter_collection.csv"):
                    f = os.path.join(root, fn)
        if f is None:
            raise SystemExit(f"no counter file under {d}")
        # the conv kernel serves the forward conv AND the input gradient: keep them apart by launch order within one iteration
        order = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            k = role(r["Kernel_Name"])
            if k is None:
                continue
            if k == "conv_or_dgrad":
                order[k] += 1
                k = "conv" if order[k] % 2 == 1 else "dgrad"
            names[k] = r["Kernel_Name"][:100]
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r["Start_Timestamp"]:
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    es = 2 if dt == "bf16" else 4
    rec = {"config": cfg, "batch": batch, "dtype": dt, "shape": f"{c} -> {c} @ {size} x {size} x {batch}", "kernels": {},
           "algorithmic_bytes_block_forward": 2.0 * c * size * size * batch * es + 9 * c * c * es,
           "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/microbench.py block; hbm_bytes = "
                   "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE counts 128-byte requests at 64 B)"}
    rows = []
    for k in sorted(per):
        m = {cn: sum(v) / len(v) for cn, v in per[k].items()}
        e = {"kernel": names[k], "avg_duration_us_under_pmc": round(sum(dur[k]) / max(1, len(dur[k])), 1)}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e.update(FETCH_SIZE_KB=m["FETCH_SIZE"], WRITE_SIZE_KB=m["WRITE_SIZE"], hbm_bytes=(2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
        rec["kernels"][k] = e
        rows.append({"role": k, **e})
    json.dump(rec, open(os.path.join(ROOT, "profiles", f"r05_pmc_canonical_block_{cfg}.json"), "w"), indent=1)
    with open(os.path.join(ROOT, "profiles", f"r05_pmc_canonical_block_{cfg}.csv"), "w", newline="") as fh:
        keys = sorted({k for r in rows for k in r})
        w = csv.DictWriter(fh, fieldnames=keys)
        w.writeheader()
        w.writerows(rows)
    for k, e in rec["kernels"].items():
        print(f"{k:16s} {e.get('avg_duration_us_under_pmc', 0):9.1f} us  hbm {e.get('hbm_bytes', 0) / 1e9:7.3f} GB  {e['kernel'][:70]}")
    print("algorithmic bytes of the forward block: %.3f GB" % (rec["algorithmic_bytes_block_forward"] / 1e9))


if __name__ == "__main__":
    main()
