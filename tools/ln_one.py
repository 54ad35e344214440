"""LayerNorm fwd / bwd on the video tower's shape (timing / sweeps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from missm_benchmark_amd import ops
rows, d = int(sys.argv[1]) if len(sys.argv) > 1 else 50432, 768
dt = torch.bfloat16
h = torch.randn(rows, d, device="cuda"); g = torch.ones(d, device="cuda"); b = torch.zeros(d, device="cuda")
y = torch.empty(rows, d, device="cuda", dtype=dt); m = torch.empty(rows, device="cuda"); r = torch.empty(rows, device="cuda")
dy = torch.randn(rows, d, device="cuda").to(dt); dh = torch.randn(rows, d, device="cuda"); dg = torch.zeros(d, device="cuda"); db = torch.zeros(d, device="cuda")
dc = torch.empty(rows, d, device="cuda", dtype=dt)
def f(): ops.layernorm_fwd(h, g, b, y, m, r, rows, d, 1e-5)
def bw(): ops.layernorm_bwd(dy, h, m, r, g, dh, dg, db, rows, d, accumulate=True, dx_cast=dc)
for fn, nm, byts in ((f, "fwd", rows * d * 6), (bw, "bwd", rows * d * 16)):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"ln {nm} rows={rows}: {us:.1f} us  {byts / us / 1e6:.2f} TB/s")
