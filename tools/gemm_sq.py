import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from missm_benchmark_amd import ops
def bench(name, fn, flops, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:44s} {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TFLOP/s", flush=True)
dt = torch.bfloat16
for (m, n, k) in ((4096, 4096, 4096), (8192, 8192, 8192), (2048, 2048, 2048), (50432, 3072, 768), (50432, 3072, 3072), (50432, 768, 8192)):
    for zero in (False, True):
        x = (torch.zeros(m, k, device="cuda") if zero else torch.randn(m, k, device="cuda")).to(dt)
        w = (torch.zeros(n, k, device="cuda") if zero else torch.randn(n, k, device="cuda")).to(dt)
        y = torch.empty(m, n, device="cuda", dtype=dt)
        bench(f"NT {m}x{n}x{k} {'zeros' if zero else 'randn'}", lambda: ops.gemm(x, w, y), 2.0 * m * n * k)
