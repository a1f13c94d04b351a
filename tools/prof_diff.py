#!/usr/bin/env python
"""Per-kernel time per step of two rocprofv3 --kernel-trace --stats runs side by side: prof_diff.py dirA dirB steps"""
import csv, sys, collections
def load(d, steps):
    rows = list(csv.DictReader(open(d + "/p_kernel_stats.csv")))
    return {r["Name"][:70]: (float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3) for r in rows}
a, b, steps = load(sys.argv[1], float(sys.argv[3])), load(sys.argv[2], float(sys.argv[3])), float(sys.argv[3])
names = sorted(set(a) | set(b), key=lambda n: -(a.get(n, (0,))[0] + b.get(n, (0,))[0]))
ta = sum(v[0] for v in a.values()); tb = sum(v[0] for v in b.values())
print(f"total ms/step: A {ta:.3f}  B {tb:.3f}")
for n in names[:30]:
    x, y = a.get(n, (0, 0, 0)), b.get(n, (0, 0, 0))
    print(f"{x[0]:8.3f} ({x[1]:5.1f} x {x[2]:7.1f} us)  {y[0]:8.3f} ({y[1]:5.1f} x {y[2]:7.1f} us)  {y[0]-x[0]:+7.3f}  {n}")
