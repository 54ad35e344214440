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
for rows in (6304, 50432):
    for (n, k) in ((3072, 768), (768, 3072), (2304, 768), (768, 768)):
        dy = torch.randn(rows, n, device="cuda").to(dt); x = torch.randn(rows, k, device="cuda").to(dt)
        dw = torch.zeros(n, k, device="cuda"); db = torch.zeros(n, device="cuda")
        res = []
        for sk in (1, 2, 3, 4, 6, 8, 12, 16, 24):
            us = bench(lambda: ops.gemm(dy, x, dw, trans_a=True, trans_b=True, splitk=sk, colsum_a=db))
            res.append(f"{sk}:{2.0*rows*n*k/us/1e6:.0f}")
        print(f"dW [{n}x{k}x{rows}] tiles={((n+127)//128)*((k+127)//128)}  TFLOP/s by splitk  " + "  ".join(res), flush=True)
