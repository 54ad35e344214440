"""Idle gaps and stream overlap of the LAST step in a rocprofv3 --kernel-trace of bench.py (multi-stream run).
Usage: trace_gaps.py <kernel_trace.csv> [window_ms]"""
import sys
import numpy as np
import pandas as pd

t = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 86e6
end = t.End_Timestamp.max()
s = t[t.Start_Timestamp >= end - win].reset_index(drop=True)
w0 = s.Start_Timestamp.iloc[0]
cur_e, last_i, gaps, busy, cur_s = s.End_Timestamp.iloc[0], 0, [], 0, s.Start_Timestamp.iloc[0]
for i in range(1, len(s)):
    a, b = s.Start_Timestamp.iloc[i], s.End_Timestamp.iloc[i]
    if a > cur_e:
        gaps.append((a - cur_e, last_i, i, cur_e - w0)); busy += cur_e - cur_s; cur_s = a
    if b > cur_e:
        cur_e, last_i = b, i
busy += cur_e - cur_s
span = s.End_Timestamp.max() - w0
print(f"window {span / 1e6:.2f} ms, union busy {busy / 1e6:.2f} ms, idle {100 * (1 - busy / span):.1f} %, kernel-time sum {(s.End_Timestamp - s.Start_Timestamp).sum() / 1e6:.2f} ms")
gaps.sort(reverse=True)
for g, li, i, at in gaps[:10]:
    print(f"  gap {g / 1e3:7.1f} us at {at / 1e6:6.2f} ms | before: {s.Kernel_Name.iloc[li][:48]} (q{s.Queue_Id.iloc[li]}) | after: {s.Kernel_Name.iloc[i][:48]} (q{s.Queue_Id.iloc[i]})")
print(f"  total gaps {sum(g[0] for g in gaps) / 1e6:.2f} ms in {len(gaps)} gaps; > 20 us: {sum(g[0] for g in gaps if g[0] > 20e3) / 1e6:.2f} ms")
for q, d in s.groupby("Queue_Id"):
    print(f"  queue {q}: {len(d)} kernels, busy {(d.End_Timestamp - d.Start_Timestamp).sum() / 1e6:.2f} ms, from {(d.Start_Timestamp.min() - w0) / 1e6:.1f} to {(d.End_Timestamp.max() - w0) / 1e6:.1f} ms")
