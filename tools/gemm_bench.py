"""GEMM micro-benchmark over the hot-path shapes (run on the GPU box): TFLOP/s per shape and layout."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from missm_benchmark_amd import ops

def bench(name, fn, flops, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:44s} {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TFLOP/s", flush=True)

def main():
    dt = torch.bfloat16
    for rows, tag in ((6304, "img"), (50432, "vid")):
        for (n, k, nm) in ((2304, 768, "qkv"), (768, 768, "out"), (3072, 768, "fc1"), (768, 3072, "fc2")):
            x = torch.randn(rows, k, device="cuda").to(dt)
            w = (torch.randn(n, k, device="cuda") * 0.02).to(dt)
            y = torch.empty(rows, n, device="cuda", dtype=dt)
            dy = torch.randn(rows, n, device="cuda").to(dt)
            dx = torch.empty(rows, k, device="cuda", dtype=dt)
            dw = torch.zeros(n, k, device="cuda")
            fl = 2.0 * rows * n * k
            bench(f"{tag} {nm} fwd NT  [{rows}x{n}x{k}]", lambda: ops.gemm(x, w, y), fl)
            bench(f"{tag} {nm} dX  NN  [{rows}x{k}x{n}]", lambda: ops.gemm(dy, w, dx, trans_b=True), fl)
            bench(f"{tag} {nm} dW  TN  [{n}x{k}x{rows}] auto-splitK", lambda: ops.gemm(dy, x, dw, trans_a=True, trans_b=True, splitk=0), fl)
            bench(f"{tag} {nm} dW  TN  [{n}x{k}x{rows}] no split", lambda: ops.gemm(dy, x, dw, trans_a=True, trans_b=True, splitk=1), fl)

if __name__ == "__main__":
    main()
