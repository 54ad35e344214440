"""Per-workgroup timeline of one TN (weight-gradient) GEMM launch: out[M,N] (fp32, split-K atomics) = a[K,M]^T b[K,N]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from missm_benchmark_amd import ops, _lib
lib = _lib.load()
M, N, K = (int(x) for x in sys.argv[1:4])
sk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dt = torch.bfloat16
mode = os.environ.get("MODE", "tn")   # tn: a[K,M] b[K,N];  nn: a[M,K] b[K,N];  nt: a[M,K] b[N,K]
ta, tb = mode[0] == "t", mode[1] == "n"
a = torch.randn((K, M) if ta else (M, K), device="cuda").to(dt); b = torch.randn((K, N) if tb else (N, K), device="cuda").to(dt)
y = torch.zeros(M, N, device="cuda")
cs = torch.zeros(M, device="cuda")
def run(): ops.gemm(a, b, y, trans_a=ta, trans_b=tb, splitk=sk, colsum_a=cs if (ta and not os.environ.get("NOCS")) else None)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 50
print(f"{mode} M{M} N{N} K{K} splitk={sk}: {us:.1f} us/launch  {2.0*M*N*K/us/1e6:.0f} TFLOP/s")
nb = 4096
dbg = torch.zeros(nb * 8, dtype=torch.int64, device="cuda")
lib.missm_gemm_set_debug_buffer(dbg.data_ptr())
run(); torch.cuda.synchronize()
lib.missm_gemm_set_debug_buffer(None)
d = dbg.cpu().numpy().reshape(nb, 8)
d = d[d[:, 0] != 0]
t0 = d[:, 0].min()
start, loop, loop_end, end = [(d[:, i] - t0) / 100.0 for i in range(4)]
print(f"blocks {len(d)}; kernel span {end.max():.1f} us")
for nm, v in (("prologue", loop - start), ("main loop", loop_end - loop), ("epilogue+drain", end - loop_end),
              ("colsum tail", (d[:, 2] - d[:, 6]) / 100.0), ("epilogue issue", (d[:, 5] - t0) / 100.0 - loop_end), ("lifetime", end - start), ("start", start)):
    print(f"  {nm:16s} median {np.median(v):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}")
