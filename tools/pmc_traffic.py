"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command) into per-kernel-family HBM-side
traffic.  FETCH_SIZE is doubled per /opt/skills/guides/MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads);
both counters are in KiB.  Usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import datetime, json, os, sys
import pandas as pd

FAMILIES = (("gemm", ("gemm_kernel", "gemm8p", "gemm4w", "splitk_reduce")), ("ln", ("ln_fwd", "ln_bwd")), ("attn", ("attn_",)), ("adam", ("adam_kernel", "adam_cast_batched")))


def family(name):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return "other"


def load(path, counter):
    d = pd.read_csv(path)
    d = d[d.Counter_Name == counter]
    d = d.assign(fam=d.Kernel_Name.map(family), is_gemm=d.Kernel_Name.str.contains("gemm_kernel|gemm8p_kernel|gemm8p_tn_kernel|gemm4w_kernel"))
    return d


STEPS = 2   # --steps 1 --warmup 1
f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
# provenance travels with the numbers (bench.py copies it into the JSON line next to the traffic figure): the commit the profiled
# tree was built from is handed in by the caller (MISSM_COMMIT=<sha> - the GPU box has no .git), the date is the pass's own
out = {"commit": os.environ.get("MISSM_COMMIT", "unknown"), "date": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"),
       "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 1 --warmup 1 "
                  "--no-cpu-baseline --no-roofline --no-fp32-line --serial-streams",
       "note": "2 steps profiled (1 warm-up + 1 timed). FETCH_SIZE in KiB, doubled per MI355X_MICROARCH.md (gfx950 reports half of wide "
               "coalesced reads); WRITE_SIZE in KiB.  gemm = gemm8p_kernel + gemm8p_tn_kernel + gemm4w_kernel + gemm_kernel + the split-K reduce kernels; its "
               "launches count the tile-kernel dispatches only (a grouped launch is one dispatch).", "kernels": {}}
out["steps_profiled"] = STEPS
for fam in sorted(set(f.fam)):
    ff, ww = f[f.fam == fam], w[w.fam == fam]
    n = int(ff.is_gemm.sum()) if fam == "gemm" else len(ff)
    fb, wb = float(ff.Counter_Value.sum()) * 1024 * 2, float(ww.Counter_Value.sum()) * 1024
    out["kernels"][fam] = {"launches": n, "fetch_GB_x2_corrected": round(fb / 1e9, 3), "write_GB": round(wb / 1e9, 3),
                           "bytes_per_launch": int((fb + wb) / max(n, 1)), "bytes_per_step": int((fb + wb) / STEPS)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
