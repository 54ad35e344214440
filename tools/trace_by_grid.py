"""Kernel-trace rows grouped by (kernel, grid size): calls, average and total time - separates the big-shape launches of a kernel
from its small ones.  Usage: trace_by_grid.py kernel_trace.csv [steps]"""
import csv, sys, collections, re
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = collections.defaultdict(lambda: [0, 0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
        a = agg[(n, grid // max(wg, 1))]
        a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
print(f"{'kernel':70s} {'blocks':>8} {'calls':>6} {'avg_us':>9} {'ms/step':>8} {'share':>6}")
for (n, g), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{n:70s} {g:8d} {a[0]:6d} {a[1]/a[0]/1e3:9.1f} {a[1]/1e6/steps:8.2f} {100*a[1]/tot:6.1f}")
print(f"total {tot/1e6/steps:.2f} ms/step")
