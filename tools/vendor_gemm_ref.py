"""Reference point only (not used by the product): what the vendor GEMM (hipBLASLt via torch.matmul) reaches on the hot-path
shapes, next to this repo's kernel - to know how much headroom the MFMA GEMM has left on each shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from missm_benchmark_amd import ops

def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

dt = torch.bfloat16
for rows in (50432, 6304):
    for n, k in ((2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304)):
        x = torch.randn(rows, k, device="cuda").to(dt); w = (torch.randn(n, k, device="cuda") * 0.02).to(dt)
        y = torch.empty(rows, n, device="cuda", dtype=dt)
        fl = 2.0 * rows * n * k
        mine = t(lambda: ops.gemm(x, w, y))
        ven = t(lambda: torch.matmul(x, w.t(), out=y))
        print(f"NT [{rows}x{n}x{k}]  ours {mine*1e3:7.1f} us {fl/mine/1e9:7.1f} TF/s | vendor {ven*1e3:7.1f} us {fl/ven/1e9:7.1f} TF/s", flush=True)
    for n, k in ((2304, 768), (3072, 768), (768, 3072), (768, 768)):
        dy = torch.randn(rows, n, device="cuda").to(dt); x = torch.randn(rows, k, device="cuda").to(dt)
        dw = torch.zeros(n, k, device="cuda"); dwb = torch.empty(n, k, device="cuda", dtype=dt)
        fl = 2.0 * rows * n * k
        mine = t(lambda: ops.gemm(dy, x, dw, trans_a=True, trans_b=True, splitk=0))
        ven = t(lambda: torch.matmul(dy.t(), x, out=dwb))
        print(f"TN [{n}x{k}x{rows}]  ours {mine*1e3:7.1f} us {fl/mine/1e9:7.1f} TF/s | vendor {ven*1e3:7.1f} us {fl/ven/1e9:7.1f} TF/s", flush=True)
