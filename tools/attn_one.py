"""Run the MFMA attention kernels on one shape (for rocprofv3 --pmc runs / timing)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from missm_benchmark_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S, H, hd = 197, 12, 64
d = H * hd
rows = N * S
dt = torch.bfloat16
qkv = torch.randn(rows, 3 * d, device="cuda").to(dt)
out = torch.empty(rows, d, device="cuda", dtype=dt)
lse = torch.empty(N * H * S, device="cuda")
dout = torch.randn(rows, d, device="cuda").to(dt)
dqkv = torch.empty(rows, 3 * d, device="cuda", dtype=dt)
for _ in range(3):
    ops.attention_fwd(qkv, out, lse, N, S, H, hd)
    ops.attention_bwd(qkv, out, dout, lse, dqkv, N, S, H, hd)
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
e0.record()
for _ in range(5): ops.attention_fwd(qkv, out, lse, N, S, H, hd)
e1.record()
for _ in range(5): ops.attention_bwd(qkv, out, dout, lse, dqkv, N, S, H, hd)
e2.record(); torch.cuda.synchronize()
fl = 4.0 * S * S * hd * N * H
print(f"N={N}: fwd {e0.elapsed_time(e1)/5*1e3:.1f} us ({fl/(e0.elapsed_time(e1)/5)/1e9:.1f} TFLOP/s)  bwd {e1.elapsed_time(e2)/5*1e3:.1f} us ({2.5*fl/(e1.elapsed_time(e2)/5)/1e9:.1f} TFLOP/s)")
