"""Attention fwd / bwd micro-benchmark at the video tower's spatial shape (256 frames x 12 heads, S = 197) and the grouped image
towers' (128 frames): time per launch, achieved algorithmic GB/s and TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from missm_benchmark_amd import ops


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


NSEQ = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else (256, 128, 32)
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for nseq in NSEQ:
    L, H, hd = int(os.environ.get("ATTN_L", "197")), 12, 64
    rows = nseq * L
    qkv = (torch.randn(rows, 3 * H * hd, device="cuda") * 0.5).to(torch.bfloat16)
    out = torch.empty(rows, H * hd, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(nseq * H * L, device="cuda")
    dout = torch.randn(rows, H * hd, device="cuda").to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    f = timed(lambda: ops.attention_fwd(qkv, out, lse, nseq, L, H, hd), ITERS)
    b = timed(lambda: ops.attention_bwd(qkv, out, dout, lse, dqkv, nseq, L, H, hd), ITERS)
    units = nseq * H
    fb, bb = units * L * hd * 4 * 2, units * L * hd * 8 * 2
    ff, bf = 4.0 * units * L * L * hd, 10.0 * units * L * L * hd
    print(f"nseq {nseq:4d}: fwd {f*1e3:7.1f} us ({fb/f/1e6:6.0f} GB/s, {ff/f/1e9:6.1f} TFLOP/s, {f*1e3/units*256:5.2f} us per unit-CU)   "
          f"bwd {b*1e3:7.1f} us ({bb/b/1e6:6.0f} GB/s, {bf/b/1e9:6.1f} TFLOP/s, {b*1e3/units*256:5.2f} us per unit-CU)", flush=True)

# time attention of the video tower: B x 197 sequences of T = 8 tokens, 197 rows apart in the [B * 8 * 197, 2304] QKV matrix
for B in ((32, 16) if os.environ.get("ATTN_TIME", "1") != "0" else ()):
    S, T, H, hd = 197, 8, 12, 64
    rows, nseq = B * T * S, B * S
    qkv = (torch.randn(rows, 3 * H * hd, device="cuda") * 0.5).to(torch.bfloat16)
    out = torch.empty(rows, H * hd, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(nseq * H * T, device="cuda")
    dout = torch.randn(rows, H * hd, device="cuda").to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    kw = dict(seq_div=S, seq_outer=T * S, seq_inner=1, tok_stride=S)
    f = timed(lambda: ops.attention_fwd(qkv, out, lse, nseq, T, H, hd, **kw), ITERS)
    b = timed(lambda: ops.attention_bwd(qkv, out, dout, lse, dqkv, nseq, T, H, hd, **kw), ITERS)
    units = nseq * H
    fb, bb = units * T * hd * 4 * 2, units * T * hd * 8 * 2
    print(f"time attention B {B:3d} ({units} units of 8 tokens): fwd {f*1e3:7.1f} us ({fb/f/1e6:6.0f} GB/s)   bwd {b*1e3:7.1f} us ({bb/b/1e6:6.0f} GB/s)", flush=True)
