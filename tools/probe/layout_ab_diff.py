import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
step, phase = 0, {}
per = [collections.defaultdict(list), collections.defaultdict(list)]
for r in rows:
    name = r["Kernel_Name"]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if 4 <= step < 12: per[0][name].append(d)
    elif 17 <= step < 25: per[1][name].append(d)
    if name.startswith("optim_step_kernel"): step += 1
tot = [sum(sum(v) for v in p.values()) / 8e6 for p in per]
print("kernel time per step: phase 1 %.2f ms, phase 2 %.2f ms" % tuple(tot))
diff = []
for k in per[0]:
    a, b = sum(per[0][k]) / 8e6, sum(per[1].get(k, [0])) / 8e6
    diff.append((a - b, k, a, b, len(per[0][k]) // 8))
for d, k, a, b, n in sorted(diff, reverse=True)[:14]:
    print("%-90s %3d/step  %7.3f -> %7.3f ms/step  (%+.3f)" % (k[:90], n, a, b, -d))
