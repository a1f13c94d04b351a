#!/bin/bash
# A/B of library options / env knobs on the kernel time INSIDE the training step: interleaved rocprofv3 --kernel-trace --stats runs of
# bench.py, compared on the per-kernel sums (resolves ~0.1 % where ms_per_step A/Bs drown in +-0.1 ms of run-to-run noise).
#   bash tools/in_step_ab.sh cfg3 "MIA_OPTIONS=conv_pw_s2=0" "MIA_OPTIONS=conv_pw_s2=1"     (through gpurun; two pairs)
cfg=$1; shift
A="$1"; B="$2"
cd /root/repo
for i in 1 2; do
  for arm in A B; do
    kv=$([ $arm = A ] && echo "$A" || echo "$B")
    bash tools/r4_prof_cfg.sh $cfg $kv > /dev/null 2>&1
    cp gpurun_out/r4p_${cfg}_kernel_stats.csv gpurun_out/instep_${arm}_$i.csv
  done
done
python - "$A" "$B" <<'PY'
import csv, sys
def load(f):
    rows = list(csv.DictReader(open(f)))
    calls = max(int(r["Calls"]) for r in rows if r["Name"].startswith("optim_step_kernel")) if any(r["Name"].startswith("optim_step_kernel") for r in rows) else 1
    return {r["Name"]: float(r["TotalDurationNs"]) / calls / 1e3 for r in rows}
for i in (1, 2):
    a, b = load(f"gpurun_out/instep_A_{i}.csv"), load(f"gpurun_out/instep_B_{i}.csv")
    print(f"pair {i}: all kernels  A ({sys.argv[1]}) {sum(a.values()) / 1e3:.3f} ms   B ({sys.argv[2]}) {sum(b.values()) / 1e3:.3f} ms per step")
    diff = sorted(((b.get(k, 0.0) - a.get(k, 0.0), k) for k in set(a) | set(b)), key=lambda t: -abs(t[0]))
    for d, k in diff[:6]:
        if abs(d) >= 5.0:
            print(f"    {d:+8.0f} us  {k[:110]}")
PY
