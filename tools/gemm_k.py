import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from missm_benchmark_amd import ops
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dt = torch.bfloat16
M = 50432
for N in (768, 3072):
    for K in (64, 128, 256, 512, 768, 1536, 3072):
        x = torch.randn(M, K, device="cuda").to(dt); w = torch.randn(N, K, device="cuda").to(dt)
        y = torch.empty(M, N, device="cuda", dtype=dt)
        us = bench(lambda: ops.gemm(x, w, y))
        print(f"M={M} N={N} K={K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s   out-bytes {M*N*2/1e6:.0f} MB -> {M*N*2/us/1e6:.2f} TB/s", flush=True)
