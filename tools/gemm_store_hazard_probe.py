"""Every element of one fc2-shaped NT product with bias + fp32 residual against fp32 torch, three launches: lists wrong elements by
tile / row in tile / column in tile and the terms of one of them.  Written to find the store-data hazard of DESIGN.md 4.1 (a 16-byte
buffer store whose soffset is a REGISTER may have its first data register rewritten by the next instruction: component 0 of rows
r = 0..2, lanes 12-15 of every 16, in tiles here and there - a 600-row spot check passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from missm_benchmark_amd import ops
dt = torch.bfloat16
rows, n, k = 50432, 768, 3072
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(rows, k, device="cuda", generator=g).to(dt)
w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(dt)
bias = torch.randn(n, device="cuda", generator=g)
res = torch.randn(rows, n, device="cuda", generator=g)
for it in range(3):
    y = torch.full((rows, n), float("nan"), device="cuda")
    ops.gemm(x, w, y, bias=bias, resid=res)
    torch.cuda.synchronize()
    bad_all = []
    for lo in range(0, rows, 8192):
        hi = min(rows, lo + 8192)
        ref = x[lo:hi].float() @ w.float().t() + bias + res[lo:hi]
        d = (y[lo:hi] - ref).abs()
        bad = (~(d < 0.05)).nonzero()
        if bad.numel():
            bad[:, 0] += lo
            bad_all.append(bad)
    if not bad_all:
        print("iter", it, "clean"); continue
    bad = torch.cat(bad_all).cpu()
    r, c = bad[:, 0], bad[:, 1]
    print("iter", it, "bad elements", len(bad), "nan", int(torch.isnan(y).sum()))
    tiles = sorted(set(zip((r // 256).tolist(), (c // 256).tolist())))
    print(" tiles (row, col):", tiles[:20], "n", len(tiles))
    print(" rows in tile:", sorted(set((r % 256).tolist()))[:40])
    print(" cols in tile:", sorted(set((c % 256).tolist()))[:70])
    # what is wrong: missing residual? missing bias?
    rr, cc = int(r[0]), int(c[0])
    ref0 = float(x[rr].float() @ w[cc].float())
    print(" sample", rr, cc, "got", float(y[rr, cc]), "acc", ref0, "bias", float(bias[cc]), "res", float(res[rr, cc]))
