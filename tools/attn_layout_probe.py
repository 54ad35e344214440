"""What would a head-major q/k/v layout buy?  The same attention work with every (sequence, head) unit's operands contiguous
(H = 1, row = [q | k | v] of one head: 384-byte rows) against the training layout (12 heads: a head's slice is 128 bytes of a 4.6 KB row)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from missm_benchmark_amd import ops

def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

L, hd = 197, 64
for name, nseq, H in (("training layout  (256 seq x 12 heads)", 256, 12), ("unit-contiguous  (3072 seq x 1 head)", 256 * 12, 1)):
    rows = nseq * L
    qkv = (torch.randn(rows, 3 * H * hd, device="cuda") * 0.5).to(torch.bfloat16)
    out = torch.empty(rows, H * hd, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(nseq * H * L, device="cuda")
    dout = torch.randn(rows, H * hd, device="cuda").to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    f = timed(lambda: ops.attention_fwd(qkv, out, lse, nseq, L, H, hd))
    b = timed(lambda: ops.attention_bwd(qkv, out, dout, lse, dqkv, nseq, L, H, hd))
    print(f"{name}: fwd {f:7.1f} us  bwd {b:7.1f} us", flush=True)
