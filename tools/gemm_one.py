"""Run one GEMM shape a few times (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from missm_benchmark_amd import ops
rows, n, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (50432, 3072, 768)))
mode = sys.argv[4] if len(sys.argv) > 4 else "nt"
dt = torch.bfloat16
x = torch.randn(rows, k, device="cuda").to(dt)
w = (torch.randn(n, k, device="cuda") * 0.02).to(dt)
y = torch.empty(rows, n, device="cuda", dtype=dt)
dy = torch.randn(rows, n, device="cuda").to(dt)
dw = torch.zeros(n, k, device="cuda")
for _ in range(5):
    if mode == "nt":
        ops.gemm(x, w, y)
    elif mode == "tn":
        ops.gemm(dy, x, dw, trans_a=True, trans_b=True, splitk=0)
torch.cuda.synchronize()
