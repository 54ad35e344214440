"""Small HIP-backed modules for the projection / fusion tail (all fp32, launch-bound).

Each is an ``nn.Module`` holding ordinary ``nn.Parameter``s (so ``state_dict()`` keys match the reference's
``nn.Linear`` / ``nn.LayerNorm`` children) whose forward/backward are ``libmissm_hip.so`` launches wrapped in an
``autograd.Function``.  No torch arithmetic is used.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
from torch import nn

from . import _lib, ops


def _gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise _lib.MissmError(f"{what}: runs only on an MI355X; there is no CPU fallback")


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, relu, row_code, code):
        _gpu(x, "HipLinear")
        x = x.contiguous().float()
        y = torch.empty(x.shape[0], w.shape[0], device=x.device, dtype=torch.float32)
        ops.small_linear_fwd(x, w, b, y, relu=relu, row_code=row_code, code=code)
        ctx.save_for_backward(x, w, y if relu else None, row_code)
        ctx.code, ctx.has_bias = code, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, row_code = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty(w.shape[0], device=w.device, dtype=torch.float32) if ctx.has_bias else None
        ops.small_linear_bwd(dy, x, w, dx, dw, db, relu_y=y, row_code=row_code, code=ctx.code)
        return dx, dw, db, None, None, None


class HipLinear(nn.Module):
    """nn.Linear stand-in: y = x W^T + b, optional fused ReLU, optional zeroing of rows whose code matches."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, relu: bool = False):
        super().__init__()
        if in_features % 4:
            raise ValueError("in_features must be a multiple of 4")
        self.in_features, self.out_features, self.relu = in_features, out_features, relu
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(self.in_features)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, row_code: Optional[torch.Tensor] = None, code: int = 0):
        return _LinearFn.apply(x, self.weight, self.bias, self.relu, row_code, code)


class _SumLinearFn(torch.autograd.Function):
    """z = sum_m where(missing == code_m, 0, e_m W_m^T + b_m): the cross-modal fusion projection in one pass."""

    @staticmethod
    def forward(ctx, missing, codes, n, *tensors):
        xs, ws, bs = tensors[:n], tensors[n:2 * n], tensors[2 * n:]
        _gpu(xs[0], "fusion projection")
        xs = tuple(x.contiguous().float() for x in xs)
        z = torch.empty(xs[0].shape[0], ws[0].shape[0], device=xs[0].device, dtype=torch.float32)
        for i in range(n):
            ops.small_linear_fwd(xs[i], ws[i], bs[i], z, row_code=missing, code=codes[i], accumulate=i > 0)
        ctx.save_for_backward(missing, *xs, *ws)
        ctx.codes, ctx.n = codes, n
        return z

    @staticmethod
    def backward(ctx, dz):
        n = ctx.n
        missing = ctx.saved_tensors[0]
        xs, ws = ctx.saved_tensors[1:1 + n], ctx.saved_tensors[1 + n:]
        dz = dz.contiguous()
        dxs, dws, dbs = [], [], []
        for i in range(n):
            dx, dw = torch.empty_like(xs[i]), torch.empty_like(ws[i])
            db = torch.empty(ws[i].shape[0], device=dz.device, dtype=torch.float32)
            ops.small_linear_bwd(dz, xs[i], ws[i], dx, dw, db, row_code=missing, code=ctx.codes[i])
            dxs.append(dx); dws.append(dw); dbs.append(db)
        return (None, None, None, *dxs, *dws, *dbs)


def fused_modal_sum(missing_index: torch.Tensor, codes: Sequence[int], xs, linears: Sequence[HipLinear]):
    n = len(xs)
    return _SumLinearFn.apply(missing_index.contiguous(), tuple(int(c) for c in codes), n, *xs, *[l.weight for l in linears],
                              *[l.bias for l in linears])


class _ConcatLinearFn(torch.autograd.Function):
    """z = cat_m(x_m W_m^T + b_m) along the feature axis, each projection writing its own column slice of z (no torch.cat);
    rows whose missing code matches modality m take the substitute input row sub_m instead of x_m (the reference overwrites
    the embedding rows with the zero / mean / median statistics buffer, src/model/baseline.py:80-82), so they still feed
    dW / db but pass no gradient back to the encoder."""

    @staticmethod
    def forward(ctx, missing, codes, n, *tensors):
        xs, ws, bs, subs = tensors[:n], tensors[n:2 * n], tensors[2 * n:3 * n], tensors[3 * n:]
        _gpu(xs[0], "fusion projection")
        xs = tuple(x.contiguous().float() for x in xs)
        F = ws[0].shape[0]
        z = torch.empty(xs[0].shape[0], n * F, device=xs[0].device, dtype=torch.float32)
        for i in range(n):
            ops.small_linear_fwd(xs[i], ws[i], bs[i], z[:, i * F:(i + 1) * F], row_code=missing if subs[i] is not None else None,
                                 code=codes[i], x_sub=subs[i])
        ctx.save_for_backward(missing, *xs, *ws, *[s if s is not None else xs[0].new_empty(0) for s in subs])
        ctx.codes, ctx.n, ctx.has_sub = codes, n, tuple(s is not None for s in subs)
        return z

    @staticmethod
    def backward(ctx, dz):
        n = ctx.n
        missing = ctx.saved_tensors[0]
        xs, ws, subs = ctx.saved_tensors[1:1 + n], ctx.saved_tensors[1 + n:1 + 2 * n], ctx.saved_tensors[1 + 2 * n:]
        dz = dz.contiguous()
        F = ws[0].shape[0]
        dxs, dws, dbs = [], [], []
        for i in range(n):
            dx, dw = torch.empty_like(xs[i]), torch.empty_like(ws[i])
            db = torch.empty(F, device=dz.device, dtype=torch.float32)
            ops.small_linear_bwd(dz[:, i * F:(i + 1) * F], xs[i], ws[i], dx, dw, db, row_code=missing if ctx.has_sub[i] else None,
                                 code=ctx.codes[i], x_sub=subs[i] if ctx.has_sub[i] else None)
            dxs.append(dx); dws.append(dw); dbs.append(db)
        return (None, None, None, *dxs, *dws, *dbs, *([None] * n))


def fused_modal_concat(missing_index, codes: Sequence[int], xs, linears: Sequence[HipLinear], substitutes=None):
    """substitutes: per modality a [feature_dims] fp32 row (imputation statistics) or None (no masking at all)."""
    n = len(xs)
    subs = list(substitutes) if substitutes is not None else [None] * n
    return _ConcatLinearFn.apply(missing_index.contiguous(), tuple(int(c) for c in codes), n, *xs, *[l.weight for l in linears],
                                 *[l.bias for l in linears], *subs)


class _IntraAttentionFn(torch.autograd.Function):
    """Intra-modality (channel) attention fusion, reference src/model/baseline.py:183-205:
         d_m = x_m W_m^T + b_m ;  g_m = sigmoid(W2 relu(W1 [d_m | r] + b1) + b2) ;  z = sum_m (missing_m ? 0 : d_m * g_m)
    with r the learned fusion representation row and W1, W2 shared by the modalities.  Per modality the projection writes
    d_m straight into the left half of the [d_m | r] row, the gate kernel applies sigmoid, product, missing-row mask and the
    sum over modalities in one pass; the backward mirrors it and accumulates the shared layers' gradients in the kernels."""

    @staticmethod
    def forward(ctx, missing, codes, n, rep, w1, b1, w2, b2, *tensors):
        xs, ws, bs = tensors[:n], tensors[n:2 * n], tensors[2 * n:]
        _gpu(xs[0], "intra-attention fusion")
        xs = tuple(x.contiguous().float() for x in xs)
        B, F = xs[0].shape[0], ws[0].shape[0]
        dev = xs[0].device
        z = torch.empty(B, F, device=dev, dtype=torch.float32)
        saved = []
        for i in range(n):
            c = torch.empty(B, 2 * F, device=dev, dtype=torch.float32)
            ops.small_linear_fwd(xs[i], ws[i], bs[i], c[:, :F])
            c[:, F:].copy_(rep.expand(B, F))                      # broadcast copy of the representation row (no arithmetic)
            h1 = torch.empty(B, w1.shape[0], device=dev, dtype=torch.float32)
            ops.small_linear_fwd(c, w1, b1, h1, relu=True)
            pre = torch.empty(B, F, device=dev, dtype=torch.float32)
            ops.small_linear_fwd(h1, w2, b2, pre)
            ops.gate_fwd(c[:, :F], pre, z, row_code=missing, code=codes[i], accumulate=i > 0)
            saved += [c, h1, pre]
        ctx.save_for_backward(missing, w1, w2, *xs, *ws, *saved)
        ctx.codes, ctx.n = codes, n
        return z

    @staticmethod
    def backward(ctx, dz):
        n, codes = ctx.n, ctx.codes
        missing, w1, w2 = ctx.saved_tensors[:3]
        xs, ws = ctx.saved_tensors[3:3 + n], ctx.saved_tensors[3 + n:3 + 2 * n]
        saved = ctx.saved_tensors[3 + 2 * n:]
        dz = dz.contiguous()
        B, F = dz.shape
        dev = dz.device
        f32 = dict(device=dev, dtype=torch.float32)
        dw1, db1 = torch.empty_like(w1), torch.empty(w1.shape[0], **f32)
        dw2, db2 = torch.empty_like(w2), torch.empty(w2.shape[0], **f32)
        drep = torch.zeros(F, **f32)
        dxs, dws, dbs = [], [], []
        for i in range(n):
            c, h1, pre = saved[3 * i:3 * i + 3]
            dc = torch.empty(B, 2 * F, **f32)
            dpre, dh1 = torch.empty(B, F, **f32), torch.empty_like(h1)
            tmp = torch.empty(B, F, **f32)
            ops.gate_bwd(dz, c[:, :F], pre, tmp, dpre, row_code=missing, code=codes[i])         # tmp = dz * g (masked)
            ops.small_linear_bwd(dpre, h1, w2, dh1, dw2, db2, accumulate_dw=i > 0)
            ops.small_linear_bwd(dh1, c, w1, dc, dw1, db1, relu_y=h1, accumulate_dw=i > 0)
            ops.gate_bwd(dz, c[:, :F], pre, dc[:, :F], dpre, row_code=missing, code=codes[i], accumulate_dd=True)   # + dz * g
            ops.colsum(dc[:, F:], drep)
            dx, dw, db = torch.empty_like(xs[i]), torch.empty_like(ws[i]), torch.empty(F, **f32)
            ops.small_linear_bwd(dc[:, :F], xs[i], ws[i], dx, dw, db)
            dxs.append(dx); dws.append(dw); dbs.append(db)
        return (None, None, None, drep.view(1, F), dw1, db1, dw2, db2, *dxs, *dws, *dbs)


def fused_intra_attention(missing_index, codes: Sequence[int], xs, linears: Sequence[HipLinear], representation, lin1: HipLinear,
                          lin2: HipLinear):
    n = len(xs)
    return _IntraAttentionFn.apply(missing_index.contiguous(), tuple(int(c) for c in codes), n, representation, lin1.weight, lin1.bias,
                                   lin2.weight, lin2.bias, *xs, *[l.weight for l in linears], *[l.bias for l in linears])


class _InterAttentionFn(torch.autograd.Function):
    """``modal_inter_attention`` core (reference src/model/baseline.py:221-234): a learned query token attends over the M projected
    modality tokens through ``nn.MultiheadAttention`` with the missing modalities as ``key_padding_mask``.  Composed from the
    library's kernels as SELF-attention over L = M + 1 positions [query | tokens]: the packed in-projection runs on all
    positions, position 0's own key is masked out together with the missing modalities, and only position 0's output is
    used - its gradient is the only non-zero upstream gradient, so every parameter gradient equals the reference's."""

    @staticmethod
    def forward(ctx, missing, codes, n, heads, query, in_w, in_b, out_w, out_b, *tensors):
        xs, ws, bs = tensors[:n], tensors[n:2 * n], tensors[2 * n:]
        B, D = xs[0].shape[0], ws[0].shape[0]
        L = n + 1
        dev = xs[0].device
        X = torch.empty(B, L * D, device=dev, dtype=torch.float32)
        X[:, :D] = query.reshape(1, D)                                  # (copy: position 0 of every sample is the query token)
        for i in range(n):
            ops.small_linear_fwd(xs[i].contiguous(), ws[i], bs[i], X[:, (1 + i) * D:(2 + i) * D])
        Xf = X.view(B * L, D)
        qkv = torch.empty(B * L, 3 * D, device=dev, dtype=torch.float32)
        ops.small_linear_fwd(Xf, in_w, in_b, qkv)
        km = torch.ones(B, L, device=dev, dtype=torch.int32)             # 1 = key may be attended (index bookkeeping on the codes)
        km[:, 0] = 0
        for i in range(n):
            km[:, 1 + i] = (missing != codes[i]).to(torch.int32)
        att = torch.empty(B * L, D, device=dev, dtype=torch.float32)
        lse = torch.empty(B * heads * L, device=dev, dtype=torch.float32)
        ops.attention_fwd(qkv, att, lse, B, L, heads, D // heads, key_mask=km)
        c0 = att.view(B, L * D)[:, :D].contiguous()
        y = torch.empty(B, D, device=dev, dtype=torch.float32)
        ops.small_linear_fwd(c0, out_w, out_b, y)
        ctx.save_for_backward(missing, query, in_w, out_w, X, qkv, att, lse, km, c0, *xs, *ws)
        ctx.meta = (codes, n, heads, B, D, L)
        return y

    @staticmethod
    def backward(ctx, dy):
        codes, n, heads, B, D, L = ctx.meta
        missing, query, in_w, out_w, X, qkv, att, lse, km, c0 = ctx.saved_tensors[:10]
        xs, ws = ctx.saved_tensors[10:10 + n], ctx.saved_tensors[10 + n:]
        dev = dy.device
        f32 = dict(device=dev, dtype=torch.float32)
        dy = dy.contiguous()
        dc0, dw_out, db_out = torch.empty(B, D, **f32), torch.empty_like(out_w), torch.empty(D, **f32)
        ops.small_linear_bwd(dy, c0, out_w, dc0, dw_out, db_out)
        datt = torch.zeros(B, L * D, **f32)
        datt[:, :D] = dc0
        dqkv = torch.empty_like(qkv)
        ops.attention_bwd(qkv, att, datt.view(B * L, D), lse, dqkv, B, L, heads, D // heads, key_mask=km)
        dX, dw_in, db_in = torch.empty(B * L, D, **f32), torch.empty_like(in_w), torch.empty(3 * D, **f32)
        ops.small_linear_bwd(dqkv, X.view(B * L, D), in_w, dX, dw_in, db_in)
        dXv = dX.view(B, L * D)
        dquery = torch.zeros(D, **f32)
        ops.colsum(dXv[:, :D], dquery, R=B)
        out = [None, None, None, None, dquery.view_as(query), dw_in, db_in, dw_out, db_out]
        dxs, dws, dbs = [], [], []
        for i in range(n):
            dx = torch.empty_like(xs[i])
            dw, db = torch.empty_like(ws[i]), torch.empty(D, **f32)
            ops.small_linear_bwd(dXv[:, (1 + i) * D:(2 + i) * D], xs[i].contiguous(), ws[i], dx, dw, db)
            dxs.append(dx); dws.append(dw); dbs.append(db)
        return tuple(out + dxs + dws + dbs)


def fused_inter_attention(missing_index, codes: Sequence[int], xs, linears: Sequence[HipLinear], query_token, in_proj_weight, in_proj_bias,
                          out_proj: HipLinear, num_heads: int):
    n = len(xs)
    return _InterAttentionFn.apply(missing_index.contiguous(), tuple(int(c) for c in codes), n, int(num_heads), query_token, in_proj_weight,
                                   in_proj_bias, out_proj.weight, out_proj.bias, *xs, *[l.weight for l in linears], *[l.bias for l in linears])


class _DedicatedDnnFn(torch.autograd.Function):
    """Dedicated-network fusion, reference src/model/baseline.py:333-353: z = full(cat_m x_m); the rows whose modality m is
    missing are overwritten by dedicated_m(cat of the OTHER modalities).  `sel[b]` = 0 (full network) or m + 1.  Each network
    runs in select mode on its own rows only (forward and backward); its input gradient is dense over its own concatenation,
    and the per-modality slices of the networks' input gradients are summed with a strided block add."""

    @staticmethod
    def forward(ctx, sel, n, w_full, b_full, *tensors):
        xs, wds, bds = tensors[:n], tensors[n:2 * n], tensors[2 * n:]
        _gpu(xs[0], "dedicated-network fusion")
        xs = tuple(x.contiguous().float() for x in xs)
        B, C = xs[0].shape
        F = w_full.shape[0]
        feat = torch.cat(xs, dim=-1)                                           # copies (no arithmetic)
        wo = [torch.cat([xs[j] for j in range(n) if j != i], dim=-1) for i in range(n)]
        z = torch.zeros(B, F, device=feat.device, dtype=torch.float32)
        ops.small_linear_fwd(feat, w_full, b_full, z, row_code=sel, code=0, select=True)
        for i in range(n):
            ops.small_linear_fwd(wo[i], wds[i], bds[i], z, row_code=sel, code=i + 1, select=True)
        ctx.save_for_backward(sel, feat, w_full, *wo, *wds)
        ctx.n, ctx.C = n, C
        return z

    @staticmethod
    def backward(ctx, dz):
        n, C = ctx.n, ctx.C
        sel, feat, w_full = ctx.saved_tensors[:3]
        wo, wds = ctx.saved_tensors[3:3 + n], ctx.saved_tensors[3 + n:]
        dz = dz.contiguous()
        f32 = dict(device=dz.device, dtype=torch.float32)
        dfeat = torch.empty_like(feat)
        dwf, dbf = torch.empty_like(w_full), torch.empty(w_full.shape[0], **f32)
        ops.small_linear_bwd(dz, feat, w_full, dfeat, dwf, dbf, row_code=sel, code=0, select=True)
        dwds, dbds = [], []
        for i in range(n):
            d_i = torch.empty_like(wo[i])
            dw, db = torch.empty_like(wds[i]), torch.empty(wds[i].shape[0], **f32)
            ops.small_linear_bwd(dz, wo[i], wds[i], d_i, dw, db, row_code=sel, code=i + 1, select=True)
            if i > 0:                                    # modalities 0..i-1 sit at the same columns in both layouts
                ops.add_block(dfeat[:, :i * C], d_i[:, :i * C])
            if i < n - 1:                                # modalities i+1..n-1 are shifted left by one block in wo[i]
                ops.add_block(dfeat[:, (i + 1) * C:], d_i[:, i * C:])
            dwds.append(dw); dbds.append(db)
        dxs = [dfeat[:, m * C:(m + 1) * C].contiguous() for m in range(n)]
        return (None, None, dwf, dbf, *dxs, *dwds, *dbds)


def fused_dedicated_dnn(selector, xs, full: HipLinear, dedicated: Sequence[HipLinear]):
    n = len(xs)
    return _DedicatedDnnFn.apply(selector.contiguous(), n, full.weight, full.bias, *xs, *[l.weight for l in dedicated],
                                 *[l.bias for l in dedicated])


class _RegressionConcatFn(torch.autograd.Function):
    """Direct-to-task representation generation, reference src/model/baseline.py:93-161: every modality is projected; where
    modality t is missing its projection is replaced by the mean of the cross-modal regressors reg_{s->t}(x_s) over the other
    modalities s (one missing code per sample, so all of them are present there); the M blocks are concatenated.
    Block t of the output row is written by proj_t with the missing rows zeroed, then the regressors add alpha = 1/(M-1) of
    their prediction to exactly those rows (select mode).  regs[t][s] is the regressor s -> t (None on the diagonal)."""

    @staticmethod
    def forward(ctx, missing, codes, n, *tensors):
        xs, wp, bp = tensors[:n], tensors[n:2 * n], tensors[2 * n:3 * n]
        rw, rb = tensors[3 * n:3 * n + n * n], tensors[3 * n + n * n:]
        _gpu(xs[0], "regression fusion")
        xs = tuple(x.contiguous().float() for x in xs)
        B, F = xs[0].shape[0], wp[0].shape[0]
        z = torch.empty(B, n * F, device=xs[0].device, dtype=torch.float32)
        alpha = 1.0 / (n - 1)
        for t in range(n):
            zt = z[:, t * F:(t + 1) * F]
            ops.small_linear_fwd(xs[t], wp[t], bp[t], zt, row_code=missing, code=codes[t])
            for s in range(n):
                if s != t:
                    ops.small_linear_fwd(xs[s], rw[t * n + s], rb[t * n + s], zt, row_code=missing, code=codes[t], select=True,
                                         alpha=alpha, accumulate=True)
        ctx.save_for_backward(missing, *xs, *wp, *[w for w in rw if w is not None])
        ctx.codes, ctx.n = codes, n
        return z

    @staticmethod
    def backward(ctx, dz):
        n, codes = ctx.n, ctx.codes
        missing = ctx.saved_tensors[0]
        xs, wp = ctx.saved_tensors[1:1 + n], ctx.saved_tensors[1 + n:1 + 2 * n]
        it = iter(ctx.saved_tensors[1 + 2 * n:])
        rw = [None if (i // n) == (i % n) else next(it) for i in range(n * n)]
        dz = dz.contiguous()
        F = wp[0].shape[0]
        f32 = dict(device=dz.device, dtype=torch.float32)
        alpha = 1.0 / (n - 1)
        dxs = [torch.empty_like(x) for x in xs]
        touched = [False] * n
        dwp, dbp, drw, drb = [], [], [None] * (n * n), [None] * (n * n)
        for t in range(n):
            dzt = dz[:, t * F:(t + 1) * F]
            dw, db = torch.empty_like(wp[t]), torch.empty(F, **f32)
            ops.small_linear_bwd(dzt, xs[t], wp[t], dxs[t], dw, db, row_code=missing, code=codes[t], accumulate_dx=touched[t])
            touched[t] = True
            dwp.append(dw); dbp.append(db)
            for s in range(n):
                if s == t:
                    continue
                w = rw[t * n + s]
                dw, db = torch.empty_like(w), torch.empty(F, **f32)
                ops.small_linear_bwd(dzt, xs[s], w, dxs[s], dw, db, row_code=missing, code=codes[t], select=True, alpha=alpha,
                                     accumulate_dx=touched[s])
                touched[s] = True
                drw[t * n + s], drb[t * n + s] = dw, db
        return (None, None, None, *dxs, *dwp, *dbp, *drw, *drb)


def fused_regression_concat(missing_index, codes: Sequence[int], xs, projs: Sequence[HipLinear], regs):
    """regs[t][s]: HipLinear of the regressor s -> t, None for s == t"""
    n = len(xs)
    flat = [regs[t][s] for t in range(n) for s in range(n)]
    return _RegressionConcatFn.apply(missing_index.contiguous(), tuple(int(c) for c in codes), n, *xs, *[l.weight for l in projs],
                                     *[l.bias for l in projs], *[None if l is None else l.weight for l in flat],
                                     *[None if l is None else l.bias for l in flat])


class _MaskedConcatFn(torch.autograd.Function):
    """features = cat_m where(missing == code_m, 0, x_m)  (reference src/model/baseline.py:370-374)"""

    @staticmethod
    def forward(ctx, missing, codes, *xs):
        _gpu(xs[0], "masked concat")
        xs = tuple(x.contiguous().float() for x in xs)
        B, C = xs[0].shape
        out = torch.empty(B, len(xs) * C, device=xs[0].device, dtype=torch.float32)
        for i, x in enumerate(xs):
            ops.masked_copy_block(out[:, i * C:(i + 1) * C], x, missing, codes[i])
        ctx.save_for_backward(missing)
        ctx.codes, ctx.C = codes, C
        return out

    @staticmethod
    def backward(ctx, dout):
        missing, = ctx.saved_tensors
        dout = dout.contiguous()
        C = ctx.C
        dxs = []
        for i, code in enumerate(ctx.codes):
            dx = torch.empty(dout.shape[0], C, device=dout.device, dtype=torch.float32)
            ops.masked_copy_block(dx, dout[:, i * C:(i + 1) * C], missing, code)
            dxs.append(dx)
        return (None, None, *dxs)


def masked_concat(missing_index, codes: Sequence[int], xs):
    return _MaskedConcatFn.apply(missing_index.contiguous(), tuple(int(c) for c in codes), *xs)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        _gpu(x, "HipLayerNorm")
        x = x.contiguous().float()
        rows, cols = x.shape
        y = torch.empty_like(x)
        mean, rstd = torch.empty(rows, device=x.device), torch.empty(rows, device=x.device)
        ops.layernorm_fwd(x, w, b, y, mean, rstd, rows, cols, eps)
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        rows, cols = x.shape
        dx = torch.empty_like(x)
        dw, db = torch.zeros_like(w), torch.zeros_like(w)
        ops.layernorm_bwd(dy.contiguous(), x, mean, rstd, w, dx, dw, db, rows, cols, accumulate=False)
        return dx, dw, db, None


class HipLayerNorm(nn.Module):
    def __init__(self, normalized_shape: int, eps: float = 1e-5):
        super().__init__()
        if normalized_shape % 4 or normalized_shape > 2048:
            raise ValueError("normalized_shape must be a multiple of 4 and <= 2048")
        self.normalized_shape, self.eps = normalized_shape, eps
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))

    def forward(self, x):
        return _LayerNormFn.apply(x, self.weight, self.bias, self.eps)


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x = x.contiguous()
        y = torch.empty_like(x)
        mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        ops.dropout_fwd(x, y, mask, p, seed)
        ctx.save_for_backward(mask)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dx = torch.empty_like(dy)
        ops.dropout_bwd(dy.contiguous(), mask, dx, ctx.p)
        return dx, None, None


class HipDropout(nn.Module):
    """nn.Dropout stand-in (src/model/baseline.py:34) with a counter-based generator seeded from torch's RNG."""

    def __init__(self, p: float = 0.1):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        return _DropoutFn.apply(x, self.p, seed)


class _L2NormScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        x = x.contiguous()
        y = torch.empty_like(x)
        ops.l2norm_scale_fwd(x, y, scale)
        ctx.save_for_backward(x)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        ops.l2norm_scale_bwd(dy.contiguous(), x, dx, ctx.scale)
        return dx, None


def l2norm_scale(x: torch.Tensor, scale: float) -> torch.Tensor:
    return _L2NormScaleFn.apply(x, float(scale))


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        _gpu(logits, "HipCrossEntropyLoss")
        logits = logits.contiguous().float()
        loss = torch.empty(1, device=logits.device, dtype=torch.float32)
        dl = torch.empty_like(logits)
        ops.cross_entropy(logits, labels.contiguous(), loss, dl)
        ctx.save_for_backward(dl)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None


class HipCrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() (mean reduction, train_ddp.py:88): loss and d(logits) from one launch."""

    def forward(self, logits, labels):
        return _CrossEntropyFn.apply(logits, labels)


class _KLLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, student, teacher, mask, temperature):
        _gpu(student, "KL_loss")
        student, teacher = student.contiguous(), teacher.detach().contiguous()      # g_t.detach(), train_ddp.py:77
        loss = torch.empty((), device=student.device, dtype=torch.float32)
        ds = torch.empty_like(student) if ctx.needs_input_grad[0] else None
        ops.kl_loss(student, teacher, loss, ds, temperature, None if mask is None else mask.contiguous())
        ctx.save_for_backward(ds)
        return loss

    @staticmethod
    def backward(ctx, g):
        (ds,) = ctx.saved_tensors
        return (None if ds is None else ds * g), None, None, None


class HipKLLoss(nn.Module):
    """the reference's ``KL_loss`` (train_ddp.py:70-79): ``kl_div(log_softmax(g_s / T), softmax(g_t.detach() / T), 'batchmean')``.
    ``mask`` restricts it to the rows ``g_s[mask], g_t[mask]`` of the self-distillation loop (train_ddp.py:238-240) without the
    gather."""

    def __init__(self, temperature: float = 0.15):
        super().__init__()
        self.temperature = temperature

    def forward(self, g_s, g_t, mask=None):
        return _KLLossFn.apply(g_s, g_t, mask, self.temperature)


class _MSELossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _gpu(a, "MSELoss")
        a, b = a.contiguous(), b.contiguous()
        loss = torch.empty((), device=a.device, dtype=torch.float32)
        da = torch.empty_like(a)
        ops.mse_loss(a, b, loss, da)
        ctx.save_for_backward(da)
        ctx.need_b = b.requires_grad
        return loss

    @staticmethod
    def backward(ctx, g):
        (da,) = ctx.saved_tensors
        d = da * g
        return d, (-d if ctx.need_b else None)


class HipMSELoss(nn.Module):
    """``nn.MSELoss()`` (mean reduction) of the MTD student mode (train_ddp.py:84)"""

    def forward(self, a, b):
        return _MSELossFn.apply(a, b)


class _SuperGATFn(torch.autograd.Function):
    """one SuperGATConv over the batch of per-sample modality graphs (see missm_sgat_fwd): lin -> attention -> (+ bias) [-> GELU]"""

    @staticmethod
    def forward(ctx, x, w, att_l, att_r, bias, node_ok, heads, gelu):
        _gpu(x, "SuperGATConv")
        B, M = node_ok.shape
        HC = w.shape[0]
        Cc = HC // heads
        x = x.contiguous()
        xp = torch.empty(B * M, HC, device=x.device, dtype=torch.float32)
        ops.small_linear_fwd(x, w, None, xp)
        out = torch.empty(B * M, HC, device=x.device, dtype=torch.float32)
        act = torch.empty_like(out) if gelu else None
        alpha = torch.empty(B * heads * M * M, device=x.device, dtype=torch.float32)
        ops.sgat_fwd(xp, att_l.contiguous(), att_r.contiguous(), node_ok, out, alpha, B, M, heads, Cc, bias=bias, out_gelu=act)
        ctx.save_for_backward(x, w, att_l, att_r, node_ok, xp, alpha, out if gelu else None)
        ctx.meta = (B, M, heads, Cc, gelu)
        return act if gelu else out

    @staticmethod
    def backward(ctx, dy):
        x, w, att_l, att_r, node_ok, xp, alpha, pre = ctx.saved_tensors
        B, M, H, Cc, gelu = ctx.meta
        dev = dy.device
        dy = dy.contiguous()
        if gelu:
            dy = ops.gelu_bwd(dy, pre, torch.empty_like(dy))
        dbias = torch.zeros(H * Cc, device=dev, dtype=torch.float32)
        ops.colsum(dy, dbias, R=B * M)
        dxp = torch.empty_like(xp)
        dlp, drp = torch.empty(B, H * Cc, device=dev, dtype=torch.float32), torch.empty(B, H * Cc, device=dev, dtype=torch.float32)
        ops.sgat_bwd(xp, att_l.contiguous(), att_r.contiguous(), node_ok, alpha, dy, dxp, dlp, drp, B, M, H, Cc)
        dl, dr = torch.zeros(H * Cc, device=dev, dtype=torch.float32), torch.zeros(H * Cc, device=dev, dtype=torch.float32)
        ops.colsum(dlp, dl, R=B)
        ops.colsum(drp, dr, R=B)
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        ops.small_linear_bwd(dxp, x, w, dx, dw, None)
        return dx, dw, dl.view_as(att_l), dr.view_as(att_r), dbias, None, None, None


class HipSuperGATConv(nn.Module):
    """``torch_geometric.nn.SuperGATConv(in, out, heads, concat)`` as the reference's fusion_gcn uses it (src/model/baseline.py:14-15),
    on dense per-sample graphs; parameter names as in torch_geometric 2.x (``lin.weight``, ``att_l``, ``att_r``, ``bias``).
    PARITY UNPINNED (torch_geometric is absent here and unpinned upstream): restated from the published 'MX' attention."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = True):
        super().__init__()
        if not concat and heads != 1:
            raise NotImplementedError("SuperGATConv(concat=False) is used with one head by the reference; the head mean is not implemented")
        self.heads, self.out_channels, self.concat = heads, out_channels, concat
        lin = nn.Module()
        lin.weight = nn.Parameter(torch.empty(heads * out_channels, in_channels))
        self.lin = lin
        self.att_l = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_r = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels))
        for p in (self.lin.weight, self.att_l, self.att_r):     # glorot, as torch_geometric initialises them
            nn.init.xavier_uniform_(p)

    def forward(self, x, node_ok, gelu: bool = False):
        return _SuperGATFn.apply(x, self.lin.weight, self.att_l, self.att_r, self.bias, node_ok.contiguous(), self.heads, gelu)


class HipFusionGCN(nn.Module):
    """``fusion_gcn`` (src/model/baseline.py:11-24): SuperGAT(in, hidden, 4 heads, concat) -> GELU -> SuperGAT(4 hidden, out, 1 head)"""

    def __init__(self, in_channels=256, hidden_dim=128, output_dim=256, heads=4):
        super().__init__()
        self.gat1 = HipSuperGATConv(in_channels, hidden_dim, heads=heads, concat=True)
        self.gat2 = HipSuperGATConv(hidden_dim * heads, output_dim, heads=1, concat=False)

    def forward(self, x, node_ok):
        """x [B * M, in] (M nodes per sample), node_ok bool [B, M] -> [B * M, out]"""
        return self.gat2(self.gat1(x, node_ok, gelu=True), node_ok)


class _StackFn(torch.autograd.Function):
    """[B, M * D] row of the modalities' [B, D] blocks (torch.stack(features, dim=1) of the graph heads, src/model/baseline.py:261,300)"""

    @staticmethod
    def forward(ctx, *xs):
        _gpu(xs[0], "stack")
        B, D = xs[0].shape
        out = torch.empty(B, len(xs) * D, device=xs[0].device, dtype=torch.float32)
        for i, x in enumerate(xs):
            ops.masked_copy_block(out[:, i * D:(i + 1) * D], x.contiguous().float(), None, 0)
        ctx.D = D
        return out

    @staticmethod
    def backward(ctx, dout):
        dout, D = dout.contiguous(), ctx.D
        outs = []
        for i in range(dout.shape[1] // D):
            dx = torch.empty(dout.shape[0], D, device=dout.device, dtype=torch.float32)
            ops.masked_copy_block(dx, dout[:, i * D:(i + 1) * D], None, 0)
            outs.append(dx)
        return tuple(outs)


def stack_blocks(xs):
    return _StackFn.apply(*xs)


class _NodeMeanFn(torch.autograd.Function):
    """mean over the M nodes of every sample: [B * M, D] -> [B, D]   (.view(B, M, -1).mean(dim=-2), src/model/baseline.py:266,322)"""

    @staticmethod
    def forward(ctx, x, B, M):
        out = torch.empty(B, x.shape[1], device=x.device, dtype=torch.float32)
        ops.mean_rows(x.contiguous(), out, B, M, x.shape[1])
        ctx.M = M
        return out

    @staticmethod
    def backward(ctx, dy):
        M = ctx.M
        g = (dy * (1.0 / M)).contiguous()                       # (one [B, D] scalar multiply, like the loss's)
        dx = torch.empty(dy.shape[0], M * dy.shape[1], device=dy.device, dtype=torch.float32)
        for i in range(M):
            ops.masked_copy_block(dx[:, i * dy.shape[1]:(i + 1) * dy.shape[1]], g, None, 0)
        return dx.view(dy.shape[0] * M, dy.shape[1]), None, None


def node_mean(x, B, M):
    return _NodeMeanFn.apply(x, B, M)


class _FillMissingFn(torch.autograd.Function):
    """where(missing == code, filled, x): the unified graph head writes the completion network's output over a missing modality's
    embedding (src/model/baseline.py:310-312; the reference does it IN PLACE on the encoder output, which is not reproduced)"""

    @staticmethod
    def forward(ctx, x, filled, missing, code):
        x, filled = x.contiguous().float(), filled.contiguous()
        out = torch.empty_like(x)
        tmp = torch.empty_like(x)
        ops.masked_copy_block(out, x, missing, code)
        ops.masked_copy_block(tmp, filled, missing, code, keep_matching=True)
        ops.add_block(out, tmp)
        ctx.save_for_backward(missing)
        ctx.code = code
        return out

    @staticmethod
    def backward(ctx, dy):
        missing, = ctx.saved_tensors
        dy = dy.contiguous()
        dx, df = torch.empty_like(dy), torch.empty_like(dy)
        ops.masked_copy_block(dx, dy, missing, ctx.code)
        ops.masked_copy_block(df, dy, missing, ctx.code, keep_matching=True)
        return dx, df, None, None


def fill_missing(x, filled, missing_index, code):
    return _FillMissingFn.apply(x, filled, missing_index.contiguous(), int(code))
