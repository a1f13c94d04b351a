cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -d $R/gpurun_out/r5_trig -o p --output-format csv -- python3 $R/tools/probe/trigger_ab.py > $R/gpurun_out/r5_trig.log 2>&1
tail -3 $R/gpurun_out/r5_trig.log
ls $R/gpurun_out/r5_trig/*/ 2>/dev/null | head; f=$(find $R/gpurun_out/r5_trig -name "*counter_collection.csv" | head -1); head -3 $f | cut -c1-400
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), list(rows[0].keys()))
# per dispatch: counters in rows (one row per counter per dispatch)
disp = collections.OrderedDict()
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "start": int(r.get("Start_Timestamp", 0) or 0), "end": int(r.get("End_Timestamp", 0) or 0)})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
step = 0; ph = [collections.defaultdict(list), collections.defaultdict(list)]
for k in sorted(disp):
    d = disp[k]
    if 2 <= step < 6: ph[0][d["name"]].append(d)
    elif 8 <= step < 12: ph[1][d["name"]].append(d)
    if d["name"].startswith("optim_step_kernel"): step += 1
for name in ph[0]:
    if not any(t in name for t in ("conv_bt_kernel<2, 4, 8>", "wgrad_bf16_bt_kernel", "norm_act_fwd_stream", "conv64_persist_kernel<true")): continue
    for i in (0, 1):
        L = ph[i][name]
        if not L: continue
        dur = sum(x["end"] - x["start"] for x in L) / len(L)
        gui = sum(x.get("GRBM_GUI_ACTIVE", 0) for x in L) / len(L)
        sq = sum(x.get("SQ_BUSY_CYCLES", 0) for x in L) / len(L)
        print("%-60s phase %d: n %3d dur %8.1f us  GRBM_GUI_ACTIVE %.3e  (-> %.3f GHz if one counter)  SQ_BUSY_CYCLES %.3e" % (name[:60], i + 1, len(L), dur / 1e3, gui, gui / max(dur, 1), sq))
PY
rm -rf $R/gpurun_out/r5_trig
