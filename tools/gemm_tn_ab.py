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
    fro = float((dw - ref).norm() / ref.norm())
    good = err < 2e-3 and fro < 1e-4 and errb < 2e-3 and erra < 4e-3 and rep
    ok &= good
    print(f"      check dW (every element) worst {err:.2e} Frobenius {fro:.2e}  db {errb:.2e}  accumulate {erra:.2e}  bit-reproducible {rep}  {'OK' if good else 'FAIL'}", flush=True)
    del dy, x, dw, ref

# ---- grouped launches: the weight gradients of four shape-identical towers (6304 rows each) in ONE call; every element of every
# group's dW and bias gradient against fp32 torch, accumulate, and run-to-run bit-reproducibility (ordered split-K reduce)
gshapes = [(6304, 2304, 768, "4x img qkv"), (6304, 768, 768, "4x img out"), (6304, 3072, 768, "4x img fc1"), (6304, 768, 3072, "4x img fc2")]
if len(sys.argv) > 1:
    gshapes = [s for s in gshapes if any(a in s[3] for a in sys.argv[1:])]
G = 4
for rows, n, k, nm in gshapes:
    g = torch.Generator(device="cuda").manual_seed(3)
    dys = [torch.randn(rows, n, device="cuda", generator=g).to(dt) for _ in range(G)]
    xs = [torch.randn(rows, k, device="cuda", generator=g).to(dt) for _ in range(G)]
    dws = [torch.zeros(n, k, device="cuda") for _ in range(G)]
    dbs = [torch.zeros(n, device="cuda") for _ in range(G)]
    fl = 2.0 * rows * n * k * G
    ms = timed(lambda: ops.gemm_grouped(dys, xs, dws, trans_a=True, trans_b=True, splitk=0, K=rows))
    print(f"[{tag}] {nm:10s} dW[{G}x {n}x{k}] over {rows} rows     {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TFLOP/s", flush=True)
    for b in dbs:
        b.zero_()
    ops.gemm_grouped(dys, xs, dws, trans_a=True, trans_b=True, splitk=0, K=rows, colsum_a=dbs)
    again = [torch.zeros(n, k, device="cuda") for _ in range(G)]
    ops.gemm_grouped(dys, xs, again, trans_a=True, trans_b=True, splitk=0, K=rows)
    for i in range(G):
        ref = dys[i].float().t() @ xs[i].float()
        err = float((dws[i] - ref).abs().max() / ref.abs().max())
        fro = float((dws[i] - ref).norm() / ref.norm())
        refb = dys[i].float().sum(0)
        errb = float((dbs[i] - refb).abs().max() / refb.abs().max())
        rep = torch.equal(again[i], dws[i])
        good = err < 2e-3 and fro < 1e-4 and errb < 2e-3 and rep
        ok &= good
        print(f"      check group {i}: dW worst element {err:.2e} Frobenius {fro:.2e}  db {errb:.2e}  bit-reproducible {rep}  {'OK' if good else 'FAIL'}", flush=True)
    acc = [d.clone() for d in dws]
    ops.gemm_grouped(dys, xs, acc, trans_a=True, trans_b=True, splitk=0, K=rows, accumulate=True)
    for i in range(G):
        erra = float((acc[i] - 2 * dws[i]).abs().max() / dws[i].abs().max())
        ok &= erra < 1e-5
        print(f"      check group {i}: accumulate {erra:.2e} {'OK' if erra < 1e-5 else 'FAIL'}", flush=True)
print("ALL OK" if ok else "SOME CHECKS FAILED")
