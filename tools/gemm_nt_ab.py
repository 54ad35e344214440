"""NT GEMM on the hot-path shapes: correctness against torch.matmul (fp32 of the same bf16 operands) and time per launch
for every epilogue the towers use.  Run once per kernel choice: MISSM_GEMM_8P=0 (16-wave 256x256), 1 (8-phase, staggered),
2 (8-phase, no stagger); MISSM_GEMM_BIG=0 forces the 128x128 kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from missm_benchmark_amd import ops

tag = f"8P={os.environ.get('MISSM_GEMM_8P', 'dflt')} BIG={os.environ.get('MISSM_GEMM_BIG', 'dflt')}"
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


def check(name, got, ref, tol):
    err = float((got.float() - ref).abs().max() / ref.abs().max())
    print(f"      check {name}: rel err {err:.2e} {'OK' if err < tol else 'FAIL'}", flush=True)
    return err < tol


ok = True
shapes = [(50432, 2304, 768, "vid qkv"), (50432, 768, 768, "vid out"), (50432, 3072, 768, "vid fc1"), (50432, 768, 3072, "vid fc2"),
          (50432, 768, 2304, "vid dX qkv"), (25216, 3072, 768, "4img fc1"), (25216, 768, 3072, "4img fc2"), (6304, 3072, 768, "img fc1"),
          (4096, 4096, 4096, "4096^3")]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if any(a in s[3] for a in sys.argv[1:])]
for rows, n, k, nm in shapes:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(rows, k, device="cuda", generator=g).to(dt)
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(dt)
    bias = torch.randn(n, device="cuda", generator=g)
    fl = 2.0 * rows * n * k
    # (1) plain bf16 out + bias
    y = torch.empty(rows, n, device="cuda", dtype=dt)
    ms = timed(lambda: ops.gemm(x, w, y, bias=bias))
    print(f"[{tag}] {nm:12s} [{rows}x{n}x{k}] bias->bf16      {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TFLOP/s", flush=True)
    sl = slice(rows - 300, rows)      # the ragged bottom rows and a stripe of every column tile
    ref = x[sl].float() @ w.float().t() + bias
    ok &= check("bias", y[sl], ref, 1e-2)
    ref0 = x[:256].float() @ w.float().t() + bias
    ok &= check("bias(top)", y[:256], ref0, 1e-2)
    # (2) fp32 out + bias + residual (out_proj / fc2)
    if n <= 768:
        res = torch.randn(rows, n, device="cuda", generator=g)
        y32 = torch.empty(rows, n, device="cuda")
        ms = timed(lambda: ops.gemm(x, w, y32, bias=bias, resid=res))
        print(f"[{tag}] {nm:12s} [{rows}x{n}x{k}] bias+resid->f32 {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TFLOP/s", flush=True)
        ok &= check("resid", y32[sl], ref + res[sl], 2e-3)
    # (3) quick_gelu with the saved pre-activation (fc1) and its backward epilogue (dX of fc2)
    if n == 3072:
        a = torch.empty(rows, n, device="cuda", dtype=dt)
        u = torch.empty(rows, n, device="cuda", dtype=dt)
        ms = timed(lambda: ops.gemm(x, w, a, bias=bias, act=ops.ACT_QGELU, aux_out=u))
        print(f"[{tag}] {nm:12s} [{rows}x{n}x{k}] bias+qgelu+u    {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TFLOP/s", flush=True)
        ok &= check("u", u[sl], ref, 1e-2)
        ok &= check("qgelu", a[sl], ref * torch.sigmoid(1.702 * ref), 1e-2)
        du = torch.empty(rows, n, device="cuda", dtype=dt)
        ms = timed(lambda: ops.gemm(x, w, du, act=ops.ACT_DQGELU, aux_in=u))
        print(f"[{tag}] {nm:12s} [{rows}x{n}x{k}] dqgelu(u)       {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TFLOP/s", flush=True)
        uf = u[sl].float()
        sg = torch.sigmoid(1.702 * uf)
        ok &= check("dqgelu", du[sl], (x[sl].float() @ w.float().t()) * (sg + 1.702 * uf * sg * (1 - sg)), 1e-2)
    del x, w, y
print("ALL OK" if ok else "SOME CHECKS FAILED")
