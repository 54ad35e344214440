"""Per-shape GEMM time INSIDE the training step: joins the call log written under MISSM_GEMM_LOG with the tile-kernel rows of a
rocprofv3 --kernel-trace CSV (same process, same order).  Usage: gemm_insitu.py shapes.log kernel_trace.csv [steps]"""
import csv, sys, collections
log = [tuple(int(x) for x in l.split()) for l in open(sys.argv[1]) if l.strip()]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = []
with open(sys.argv[2]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if ("gemm8p_kernel" in n or "gemm8p_tn_kernel" in n or "gemm_kernel" in n or "gemm4w_kernel" in n) and "splitk" not in n:
            rows.append((int(r["Dispatch_Id"]), n, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
assert len(rows) == len(log), (len(rows), len(log))
agg = collections.defaultdict(lambda: [0, 0.0, ""])
for (disp, name, ns), sh in zip(rows, log):
    a = agg[sh]
    a[0] += 1; a[1] += ns
    a[2] = "8p" if "gemm8p_kernel" in name else ("8p_tn" if "gemm8p_tn" in name else ("4w" if "gemm4w" in name else "128"))
tot = sum(a[1] for a in agg.values())
print(f"{'M':>6} {'N':>5} {'K':>6} ta tb act f32 res ain aout cs acc grp kern   calls   avg_us  TFLOP/s  ms/step  share")
for sh, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, ta, tb, act, f32, res, ain, aout, cs, acc, grp = sh
    us = a[1] / a[0] / 1e3
    tf = 2.0 * M * N * K * grp / (us * 1e-6) / 1e12
    print(f"{M:6d} {N:5d} {K:6d} {ta:2d} {tb:2d} {act:3d} {f32:3d} {res:3d} {ain:3d} {aout:4d} {cs:2d} {acc:3d} {grp:3d} {a[2]:>5} {a[0]:6d} {us:8.1f} {tf:8.1f} {a[1]/1e6/steps:8.2f} {100*a[1]/tot:6.1f}")
print(f"total {tot/1e6/steps:.2f} ms/step over {steps} step(s)")
