"""Per-CU tile sequences of one persistent 8-phase GEMM launch (in-kernel stamps + HW_ID / XCC_ID): where the span goes."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from missm_benchmark_amd import ops, _lib
lib = _lib.load()
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (50432, 2304, 768)
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
d = d[d[:, 0] != 0]; nb = len(d)
t0 = d[:, 0].min()
start, loop, loop_end, end = [(d[:, i] - t0) / 100.0 for i in range(4)]
hw = d[:, 4]; xcc = d[:, 7]
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
by = collections.defaultdict(list)
for i in range(nb): by[int(cu[i])].append((start[i], loop[i], loop_end[i], end[i]))
print(f"tiles {nb}; CUs seen {len(by)}; span {end.max():.1f} us; sum of tile lifetimes / CUs = {np.sum(end - start) / len(by):.1f} us")
cnt = collections.Counter(len(v) for v in by.values())
print("tiles per CU:", dict(sorted(cnt.items())))
gaps, firsts, lasts, busy = [], [], [], []
for k, v in by.items():
    v.sort()
    firsts.append(v[0][0]); lasts.append(v[-1][3]); busy.append(sum(e - s for s, _, _, e in v))
    for a, b in zip(v, v[1:]): gaps.append(b[0] - a[3])
print(f"first tile start per CU: median {np.median(firsts):.2f} max {np.max(firsts):.2f}")
print(f"last tile end per CU: min {np.min(lasts):.1f} median {np.median(lasts):.1f} max {np.max(lasts):.1f}")
print(f"gap between consecutive tiles on a CU: median {np.median(gaps):.2f} p90 {np.percentile(gaps, 90):.2f} max {np.max(gaps):.2f}")
print(f"busy time per CU: min {np.min(busy):.1f} median {np.median(busy):.1f} max {np.max(busy):.1f}")
per = [ (loop - start), (loop_end - loop), (end - loop_end) ]
for nm, a in zip(("top", "main", "epi"), per): print(f"{nm}: mean {a.mean():.2f} median {np.median(a):.2f} p90 {np.percentile(a, 90):.2f} max {a.max():.2f}")
byx = collections.defaultdict(list)
for k, v in by.items(): byx[k >> 8].append(v[-1][3])
print("last end per XCD:", {k: round(max(v), 1) for k, v in sorted(byx.items())})
slow = sorted(by.items(), key=lambda kv: kv[1][-1][3])[-3:]
for k, v in slow: print("slowest CU", hex(k), [f"{s:.1f}-{e:.1f}" for s, _, _, e in v])
