"""Do two GEMM streams overlap each other's fill/drain bubbles?  sequential vs concurrent time for 2 x n launches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from missm_benchmark_amd import ops
dt = torch.bfloat16
M, N, K = 50432, 3072, 768
xs = [torch.randn(M, K, device="cuda").to(dt) for _ in range(2)]
ws = [torch.randn(N, K, device="cuda").to(dt) for _ in range(2)]
ys = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(2)]
dy = torch.randn(M, N, device="cuda").to(dt); dw = torch.zeros(N, K, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
n = 10
def seq():
    for i in range(n):
        ops.gemm(xs[0], ws[0], ys[0]); ops.gemm(xs[1], ws[1], ys[1])
def conc():
    with torch.cuda.stream(s1):
        for i in range(n): ops.gemm(xs[0], ws[0], ys[0])
    with torch.cuda.stream(s2):
        for i in range(n): ops.gemm(xs[1], ws[1], ys[1])
def conc_tn():
    with torch.cuda.stream(s1):
        for i in range(n): ops.gemm(xs[0], ws[0], ys[0])
    with torch.cuda.stream(s2):
        for i in range(n): ops.gemm(dy, xs[1], dw, trans_a=True, trans_b=True, splitk=0)
def seq_tn():
    for i in range(n):
        ops.gemm(xs[0], ws[0], ys[0]); ops.gemm(dy, xs[1], dw, trans_a=True, trans_b=True, splitk=0)
for name, fn in (("seq NT+NT", seq), ("conc NT|NT", conc), ("seq NT+TN", seq_tn), ("conc NT|TN", conc_tn)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"{name}: {t*1e3:.2f} ms  -> {2*n*2.0*M*N*K/t/1e12:.1f} TFLOP/s")
