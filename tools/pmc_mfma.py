"""MFMA utilisation and HBM-side bandwidth per kernel family from rocprofv3 passes of
  python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --serial-streams
(1) --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace   (2) the FETCH_SIZE / WRITE_SIZE passes (pmc_traffic.py output).
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (kernel time x 2.4 GHz x 1024); GRBM_GUI_ACTIVE / 8 XCDs
over the same kernels gives 2.4-2.5 GHz for the long kernels (it also counts dispatch overhead around short ones).
Usage: pmc_mfma.py <counter_collection.csv> <kernel_trace.csv> <traffic.json> <out.json>"""
import json, sys
import pandas as pd

d, t = pd.read_csv(sys.argv[1]), pd.read_csv(sys.argv[2])
traffic = json.load(open(sys.argv[3]))["kernels"]
t["dur"] = t.End_Timestamp - t.Start_Timestamp
p = d.pivot_table(index=["Dispatch_Id", "Kernel_Name"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
p = p.merge(t[["Dispatch_Id", "dur"]], on="Dispatch_Id")


def fam(n):
    for k, f in (("gemm_kernel", "gemm"), ("gemm8p", "gemm"), ("gemm4w", "gemm"), ("splitk_reduce", "gemm"), ("attn_", "attn"), ("ln_", "ln"), ("adam_kernel", "adam"), ("adam_cast_batched", "adam")):
        if k in n:
            return f
    return "other"


p["fam"] = p.Kernel_Name.map(fam)
out = {"note": "2 steps (1 warm-up + 1 timed), one HIP stream; utilisation over each family's own kernel time", "families": {}}
for f, g in p.groupby("fam"):
    ms = g.dur.sum() / 1e6 / 2
    rec = {"kernel_ms_per_step": round(ms, 2),
           "mfma_util_at_2.4GHz": round(float(g.SQ_VALU_MFMA_BUSY_CYCLES.sum() / (g.dur.sum() * 2.4 * 1024)), 4)}
    if f in traffic:
        rec["hbm_TB_per_s"] = round(traffic[f]["bytes_per_step"] / (ms * 1e-3) / 1e12, 2)
    out["families"][f] = rec
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
