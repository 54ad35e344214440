"""Per-workgroup timeline of one GEMM launch from in-kernel s_memrealtime stamps (diagnostic build knob)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from missm_benchmark_amd import ops, _lib
lib = _lib.load()
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (50432, 3072, 768)
dt = torch.bfloat16
x = torch.randn(M, K, device="cuda").to(dt); w = torch.randn(N, K, device="cuda").to(dt)
y = torch.empty(M, N, device="cuda", dtype=dt)
ACT = int(os.environ.get('ACT', '0'))
aux = torch.empty(M, N, device='cuda', dtype=dt) if ACT else None
def run(): ops.gemm(x, w, y, act=ACT, aux_out=aux if ACT == ops.ACT_QGELU else None, aux_in=aux if ACT == ops.ACT_DQGELU else None)
for _ in range(3): run()
nb = ((M + 127) // 128) * ((N + 127) // 128)
dbg = torch.zeros(nb * 8, dtype=torch.int64, device="cuda")
lib.missm_gemm_set_debug_buffer(dbg.data_ptr())
run(); torch.cuda.synchronize()
lib.missm_gemm_set_debug_buffer(None)
d = dbg.cpu().numpy().reshape(nb, 8)
d = d[d[:, 0] != 0]; nb = len(d)     # (the 256x256 kernel uses a quarter of the stamp slots)
t0 = d[:, 0].min()
start, loop, loop_end, end = [(d[:, i] - t0) / 100.0 for i in range(4)]   # microseconds (100 MHz)
hw = d[:, 4]
print(f"blocks {nb}; kernel span {end.max():.1f} us")
print(f"prologue (start->first tile landed): median {np.median(loop - start):.2f} us  p90 {np.percentile(loop - start, 90):.2f}")
print(f"main loop: median {np.median(loop_end - loop):.2f} us  p90 {np.percentile(loop_end - loop, 90):.2f}")
print(f"epilogue (+store drain): median {np.median(end - loop_end):.2f} us  p90 {np.percentile(end - loop_end, 90):.2f}")
print(f"epilogue issue only (before store drain): median {np.median((d[:, 5] - t0) / 100.0 - loop_end):.2f} us")
print(f"block lifetime: median {np.median(end - start):.2f} us")
# per-CU slot utilisation: group by hw id (cu/se/xcc bits), look at gaps between consecutive blocks in the same wave slot
key = hw & 0xffffffff
import collections
by = collections.defaultdict(list)
for i in range(nb): by[int(key[i])].append((start[i], end[i]))
gaps = []
for k, v in by.items():
    v.sort()
    for (s0, e0), (s1, e1) in zip(v, v[1:]): gaps.append(s1 - e0)
if gaps: print(f"same-slot relaunch gap: median {np.median(gaps):.2f} us  p90 {np.percentile(gaps, 90):.2f}  (slots seen {len(by)})")
first = np.sort(start)[:600]
print("start times of first 512 blocks: p50 %.2f p99 %.2f ; block #513 starts at %.2f" % (np.median(first[:512]), np.percentile(first[:512], 99), first[512] if nb > 512 else -1))
