"""Per-kernel, per-dispatch averages of rocprofv3 counter_collection CSVs (any number of passes)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0][:90]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(acc):
    if "attn" not in k and "gemm" not in k and len(sys.argv) < 99 and not any(x in k for x in ("ln_", "layernorm")):
        continue
    print(k)
    for c in sorted(acc[k]):
        n = len(cnt[k][c])
        print(f"    {c:32s} {acc[k][c] / n:16.0f}   ({n} dispatches)")
