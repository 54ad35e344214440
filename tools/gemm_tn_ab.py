"""Weight-gradient (TN) GEMM on the video-tower shapes: correctness against torch (fp32 of the same bf16 operands), the bias
gradient riding in the launch, accumulation, and time per launch.  MISSM_GEMM_8P_TN=0 old 128x128 split-K kernel, 1 8-phase."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from missm_benchmark_amd import ops

tag = f"8P_TN={os.environ.get('MISSM_GEMM_8P_TN', 'dflt')}"
dt = torch.bfloat16


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


ok = True
shapes = [(50432, 2304, 768, "vid qkv"), (50432, 768, 768, "vid out"), (50432, 3072, 768, "vid fc1"), (50432, 768, 3072, "vid fc2"),
          (25216, 3072, 768, "4img fc1"), (16384, 768, 768, "16k out"),
          (6304, 2304, 768, "img qkv"), (6304, 768, 768, "img out"), (6304, 3072, 768, "img fc1"), (6304, 768, 3072, "img fc2")]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if any(a in s[3] for a in sys.argv[1:])]
for rows, n, k, nm in shapes:
    g = torch.Generator(device="cuda").manual_seed(2)
    dy = torch.randn(rows, n, device="cuda", generator=g).to(dt)
    x = torch.randn(rows, k, device="cuda", generator=g).to(dt)
    dw = torch.zeros(n, k, device="cuda")
    db = torch.zeros(n, device="cuda")
    fl = 2.0 * rows * n * k
    ms = timed(lambda: ops.gemm(dy, x, dw, trans_a=True, trans_b=True, splitk=0, K=rows))
    print(f"[{tag}] {nm:10s} dW[{n}x{k}] over {rows} rows        {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TFLOP/s", flush=True)
    ms = timed(lambda: ops.gemm(dy, x, dw, trans_a=True, trans_b=True, splitk=0, K=rows, colsum_a=db))
    print(f"[{tag}] {nm:10s} dW + bias-gradient ride                {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TFLOP/s", flush=True)
    db.zero_()
    ops.gemm(dy, x, dw, trans_a=True, trans_b=True, splitk=0, K=rows, colsum_a=db)
    ref = (dy.float().t() @ x.float())
    err = float((dw - ref).abs().max() / ref.abs().max())
    refb = dy.float().sum(0)
    errb = float((db - refb).abs().max() / refb.abs().max())
    dw2 = dw.clone()
    ops.gemm(dy, x, dw2, trans_a=True, trans_b=True, splitk=0, K=rows, accumulate=True)
    erra = float((dw2 - 2 * ref).abs().max() / ref.abs().max())
    dw3 = torch.zeros_like(dw)
    ops.gemm(dy, x, dw3, trans_a=True, trans_b=True, splitk=0, K=rows)
    rep = torch.equal(dw3, dw)
    good = err < 2e-3 and errb < 2e-3 and erra < 4e-3 and rep
    ok &= good
    print(f"      check dW {err:.2e}  db {errb:.2e}  accumulate {erra:.2e}  bit-reproducible {rep}  {'OK' if good else 'FAIL'}", flush=True)
    del dy, x, dw, ref
print("ALL OK" if ok else "SOME CHECKS FAILED")
