"""Per-shape HBM-side traffic of the GEMM launches INSIDE the training step: joins the call log written under MISSM_GEMM_LOG with two
rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command, same launch order) and prints, per shape, the
fetched / written bytes per launch next to the operand bytes (A + B [+ residual / aux in] read once, C [+ aux out] written once).
FETCH_SIZE is doubled per /opt/skills/guides/MI355X_MICROARCH.md (gfx950 counts wide coalesced reads at half); both counters in KiB.
The split-K reduce kernels are attributed to the weight-gradient launch in front of them.
Usage: gemm_traffic_insitu.py shapes_fetch.log fetch_counter_collection.csv shapes_write.log write_counter_collection.csv [steps]"""
import csv, sys, collections


def load(logp, csvp, counter):
    log = [tuple(int(x) for x in l.split()) for l in open(logp) if l.strip()]
    rows = []
    with open(csvp) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    per_launch, i = [], -1
    for disp, name, val in rows:
        tile = ("gemm8p_kernel" in name or "gemm8p_tn_kernel" in name or "gemm_kernel" in name or "gemm4w_kernel" in name) and "splitk" not in name
        if tile:
            per_launch.append(val)
            i += 1
        elif "splitk_reduce" in name and i >= 0:
            per_launch[i] += val
    assert len(per_launch) == len(log), (len(per_launch), len(log))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for v, sh in zip(per_launch, log):
        agg[sh][0] += 1
        agg[sh][1] += v
    return agg


steps = int(sys.argv[5]) if len(sys.argv) > 5 else 1
fe, wr = load(sys.argv[1], sys.argv[2], "FETCH_SIZE"), load(sys.argv[3], sys.argv[4], "WRITE_SIZE")
print(f"{'M':>6} {'N':>5} {'K':>6} ta tb act f32 res ain aout grp  calls  fetch_MB  ideal_rd  x     write_MB ideal_wr  x     GB/step")
tot_f = tot_w = tot_i = 0.0
lines = []
for sh in fe:
    M, N, K, ta, tb, act, f32, res, ain, aout, cs, acc, grp = sh
    n = fe[sh][0]
    f_mb = fe[sh][1] / n * 1024 * 2 / 1e6
    w_mb = wr[sh][1] / wr[sh][0] * 1024 / 1e6
    es = 2
    rd = (M * K + N * K) * es * grp + (M * N * 4 * grp if res else 0) + (M * N * es * grp if ain else 0) + (M * N * 4 * grp if acc else 0)
    wb = M * N * (4 if f32 else es) * grp + (M * N * es * grp if aout else 0)
    gb = (f_mb + w_mb) * n / steps / 1e3
    tot_f += f_mb * n / steps; tot_w += w_mb * n / steps; tot_i += (rd + wb) / 1e6 * n / steps
    lines.append((gb, f"{M:6d} {N:5d} {K:6d} {ta:2d} {tb:2d} {act:3d} {f32:3d} {res:3d} {ain:3d} {aout:4d} {grp:3d} {n:6d} {f_mb:9.1f} {rd/1e6:9.1f} {f_mb/(rd/1e6):5.2f} {w_mb:9.1f} {wb/1e6:8.1f} {w_mb/(wb/1e6):5.2f} {gb:8.2f}"))
for _, l in sorted(lines, reverse=True):
    print(l)
print(f"total per step: fetched {tot_f/1e3:.1f} GB, written {tot_w/1e3:.1f} GB, operand bytes {tot_i/1e3:.1f} GB  ->  {(tot_f+tot_w)/tot_i:.2f}x")
