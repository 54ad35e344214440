"""Tensor-level wrappers over the C ABI: torch tensors in, device pointers + current HIP stream out.

PyTorch is plumbing here (device memory, streams); all arithmetic happens in ``libmissm_hip.so``.
Every wrapper validates shapes on the host before the launch so a kernel never sees a bad extent.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib

F32, BF16 = 0, 1
ACT_NONE, ACT_QGELU, ACT_GELU, ACT_DQGELU, ACT_DGELU, ACT_RELU = 0, 1, 2, 3, 4, 5
ACT_CODE = {"quick_gelu": ACT_QGELU, "gelu": ACT_GELU}
ACT_GRAD = {ACT_QGELU: ACT_DQGELU, ACT_GELU: ACT_DGELU}


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def torch_dtype(code: int) -> torch.dtype:
    return torch.float32 if code == F32 else torch.bfloat16


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _s():
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise _lib.MissmError(f"{name}: tensor must live on the GPU (no CPU fallback exists)")
    if not t.is_contiguous() and t.dim() > 0 and t.stride(-1) != 1:
        raise _lib.MissmError(f"{name}: innermost dimension must be contiguous")


GEMM_PROFILE = None  # bench.py sets this to a list: (start_event, end_event, flops) per launch
ATTN_PROFILE = None  # likewise for the attention launches: (start_event, end_event, flops, algorithmic bytes, 'fwd' | 'bwd')


def gemm(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, *, trans_a: bool = False, trans_b: bool = False, bias=None,
         resid=None, aux_in=None, aux_out=None, act: int = ACT_NONE, alpha: float = 1.0, accumulate: bool = False,
         splitk: int = 1, colsum_a=None, M=None, N=None, K=None):
    """out[M,N] = alpha * op(a) @ op(b) (+bias) -> act (+resid).
    a is [M,K] (or [K,M] with trans_a); b is [N,K] (or [K,N] with trans_b); out is the operand dtype or fp32.
    splitk: 1 off, 0 auto, >1 K slices (fp32 ``out``; slices meet in a library-owned workspace, no atomics)."""
    _req(a, "gemm a"); _req(b, "gemm b"); _req(out, "gemm out")
    if a.dtype != b.dtype:
        raise _lib.MissmError("gemm: operand dtypes differ")
    am, ak = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    bn, bk = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    M = am if M is None else M
    N = bn if N is None else N
    K = ak if K is None else K
    if K > ak or K > bk or M > am or N > bn or out.shape[0] < M or out.shape[1] < N:
        raise _lib.MissmError(f"gemm: shapes a{tuple(a.shape)} b{tuple(b.shape)} out{tuple(out.shape)} vs M{M} N{N} K{K}")
    out_f32 = int(out.dtype == torch.float32)
    if not out_f32 and (out.dtype != a.dtype or resid is not None or accumulate or splitk != 1):
        raise _lib.MissmError("gemm: output must be fp32 (required for resid/accumulate/split-K) or the operand dtype")
    aux = aux_in if aux_in is not None else aux_out
    ldaux = aux.stride(0) if aux is not None else out.stride(0)
    if aux is not None and aux.dtype != a.dtype:
        raise _lib.MissmError("gemm: aux dtype must match the operands")
    if resid is not None and (resid.dtype != torch.float32 or resid.stride(0) != out.stride(0)):
        raise _lib.MissmError("gemm: resid must be fp32 with the output's leading dimension")
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("missm_gemm", a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, a.stride(0), b.stride(0), out.stride(0),
              int(trans_a), int(trans_b), float(alpha), _p(bias), _p(resid), _p(aux_in), _p(aux_out), ldaux, act, out_f32,
              int(accumulate), int(splitk), _p(colsum_a), dt(a), _s())
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * N * K, (M, N, K, int(trans_a), int(trans_b), act)))
    return out


def gemm_grouped(as_, bs, outs, *, trans_a: bool = False, trans_b: bool = False, bias=None, resid=None, aux_in=None, aux_out=None,
                 act: int = ACT_NONE, alpha: float = 1.0, accumulate: bool = False, splitk: int = 1, colsum_a=None, M=None, N=None, K=None):
    """`len(as_)` products of ONE shape (the same linear of several shape-identical towers) in one call; every operand is a list.
    One tower: exactly `gemm`.  The library shares one grid between the problems where a grouped kernel covers the shape."""
    G = len(as_)
    opt = lambda lst, i: None if lst is None else lst[i]   # noqa: E731
    if G == 1:
        return [gemm(as_[0], bs[0], outs[0], trans_a=trans_a, trans_b=trans_b, bias=opt(bias, 0), resid=opt(resid, 0), aux_in=opt(aux_in, 0),
                     aux_out=opt(aux_out, 0), act=act, alpha=alpha, accumulate=accumulate, splitk=splitk, colsum_a=opt(colsum_a, 0), M=M, N=N, K=K)]
    import ctypes as C
    a, b, out = as_[0], bs[0], outs[0]
    for lst, name in ((as_, "a"), (bs, "b"), (outs, "out")):
        for t in lst:
            _req(t, "gemm_grouped " + name)
            if t.shape != lst[0].shape or t.stride(0) != lst[0].stride(0) or t.dtype != lst[0].dtype:
                raise _lib.MissmError(f"gemm_grouped: the {name} operands must share shape, leading dimension and dtype")
    for lst in (bias, resid, aux_in, aux_out, colsum_a):
        if lst is not None and (len(lst) != G or any(t is None for t in lst) or any(t.shape != lst[0].shape or t.stride() != lst[0].stride() for t in lst)):
            raise _lib.MissmError("gemm_grouped: optional operands are given for every group (one shape) or for none")
    am, ak = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    bn, bk = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    M = am if M is None else M
    N = bn if N is None else N
    K = ak if K is None else K
    if K > ak or K > bk or M > am or N > bn or out.shape[0] < M or out.shape[1] < N or a.dtype != b.dtype:
        raise _lib.MissmError(f"gemm_grouped: shapes a{tuple(a.shape)} b{tuple(b.shape)} out{tuple(out.shape)} vs M{M} N{N} K{K}")
    out_f32 = int(out.dtype == torch.float32)
    if not out_f32 and (out.dtype != a.dtype or resid is not None or accumulate or splitk != 1):
        raise _lib.MissmError("gemm_grouped: output must be fp32 (required for resid/accumulate/split-K) or the operand dtype")
    aux = aux_in if aux_in is not None else aux_out
    ldaux = aux[0].stride(0) if aux is not None else out.stride(0)
    if aux is not None and aux[0].dtype != a.dtype:
        raise _lib.MissmError("gemm_grouped: aux dtype must match the operands")
    if resid is not None and (resid[0].dtype != torch.float32 or resid[0].stride(0) != out.stride(0)):
        raise _lib.MissmError("gemm_grouped: resid must be fp32 with the output's leading dimension")
    arr = lambda lst: None if lst is None else (C.c_void_p * G)(*[t.data_ptr() for t in lst])   # noqa: E731
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("missm_gemm_grouped", G, arr(as_), arr(bs), arr(outs), M, N, K, a.stride(0), b.stride(0), out.stride(0), int(trans_a),
              int(trans_b), float(alpha), arr(bias), arr(resid), arr(aux_in), arr(aux_out), ldaux, act, out_f32, int(accumulate),
              int(splitk), arr(colsum_a), dt(a), _s())
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * N * K * G, (M, N, K, int(trans_a), int(trans_b), act, G)))
    return outs


def gemm_nt(a, b, out, **kw):
    """out[M,N] = a[M,K] @ b[N,K]^T ... (the forward-linear form; see gemm)."""
    return gemm(a, b, out, trans_a=False, trans_b=False, **kw)


def transpose_pad(x: torch.Tensor, out: torch.Tensor, colsum: Optional[torch.Tensor] = None, R=None, C=None):
    """out[C, ldo] = x[R, C]^T, zero padded on the right up to out.stride(0)."""
    R = x.shape[0] if R is None else R
    C = x.shape[1] if C is None else C
    if out.shape[0] < C or out.stride(0) < R or out.dtype != x.dtype:
        raise _lib.MissmError("transpose_pad: bad output")
    _lib.call("missm_transpose_pad", x.data_ptr(), out.data_ptr(), R, C, x.stride(0), out.stride(0), _p(colsum), dt(x), _s())
    return out


def colsum(x: torch.Tensor, out: torch.Tensor, div: int = 1, mod: int = 1, R=None):
    R = x.shape[0] if R is None else R
    Cn = x.shape[1]
    if out.numel() < mod * Cn or out.dtype != torch.float32:
        raise _lib.MissmError("colsum: bad output")
    _lib.call("missm_colsum", x.data_ptr(), out.data_ptr(), R, Cn, x.stride(0), div, mod, dt(x), _s())
    return out


def cast_weight(src: torch.Tensor, dst: Optional[torch.Tensor], dst_t: Optional[torch.Tensor]):
    R, Cn = src.shape
    ref = dst if dst is not None else dst_t
    _lib.call("missm_cast_weight", src.data_ptr(), _p(dst), _p(dst_t), R, Cn, dt(ref), _s())


def build_cast_table(entries, device):
    """entries: [(src fp32 [R,C], dst or None, dst_t or None)] -> device uint8 tensor of CastTile records + tile count"""
    import numpy as np
    rec = np.dtype([("src", "<u8"), ("dst", "<u8"), ("dst_t", "<u8"), ("R", "<i4"), ("C", "<i4"), ("r0", "<i4"), ("c0", "<i4")])
    rows = []
    for src, dst, dst_t in entries:
        R, Cn = src.shape
        for r0 in range(0, R, 64):
            for c0 in range(0, Cn, 64):
                rows.append((src.data_ptr(), 0 if dst is None else dst.data_ptr(), 0 if dst_t is None else dst_t.data_ptr(), R, Cn, r0, c0))
    arr = np.array(rows, dtype=rec)
    assert rec.itemsize == 40
    return torch.from_numpy(arr.view(np.uint8).copy()).to(device), len(rows)


def cast_weights_batched(table: torch.Tensor, ntiles: int, dtype_code: int):
    _lib.call("missm_cast_weights_batched", table.data_ptr(), ntiles, dtype_code, _s())


def layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, cols, eps, *, add=None, add_div=1, add_mod=1, in_mul=1, in_off=None):
    _lib.call("missm_layernorm_fwd", x.data_ptr(), x.data_ptr() if add is not None else None, _p(add), add_div, add_mod, in_mul,
              _p(in_off), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _p(mean), _p(rstd), rows, cols, float(eps), dt(y), _s())
    return y


def layernorm_bwd(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, rows, cols, *, accumulate=True, dy_div=1, dy_scale=1.0,
                  in_mul=1, in_off=None, dx_cast=None, gsum=None, gs_div=1, gs_mod=1):
    """gsum (fp32 [gs_mod, cols], caller zeroes): gsum[(row // gs_div) % gs_mod] += the updated dx rows, in the same pass;
    gs_div = 0: gsum[row % gs_mod] instead"""
    if dx_cast is not None and dx_cast.dtype != dy.dtype:
        raise _lib.MissmError("layernorm_bwd: dx_cast must have dy's dtype")
    if gsum is not None:
        if dy_div != 1 or in_mul != 1 or in_off is not None or gsum.dtype != torch.float32 or gsum.numel() < gs_mod * cols:
            raise _lib.MissmError("layernorm_bwd: group sums need the plain row mapping and an fp32 [gs_mod, cols] output")
        _lib.call("missm_layernorm_bwd_groupsum", dy.data_ptr(), float(dy_scale), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                  gamma.data_ptr(), dx.data_ptr(), int(accumulate), _p(dgamma), _p(dbeta), _p(dx_cast), gsum.data_ptr(), gs_div, gs_mod,
                  rows, cols, dt(dy), _s())
        return dx
    _lib.call("missm_layernorm_bwd", dy.data_ptr(), dy_div, float(dy_scale), x.data_ptr(), in_mul, _p(in_off), mean.data_ptr(),
              rstd.data_ptr(), gamma.data_ptr(), dx.data_ptr(), int(accumulate), _p(dgamma), _p(dbeta), _p(dx_cast), rows, cols,
              dt(dy), _s())
    return dx


def _ptrs(lst):
    import ctypes as C
    return (C.c_void_p * len(lst))(*[None if t is None else t.data_ptr() for t in lst])


def layernorm_fwd_lanes(xs, gammas, betas, ys, means, rstds, rows, cols, eps):
    """the same LayerNorm of several lock-step towers: one launch for two or more lanes, `layernorm_fwd` for one"""
    if len(xs) == 1:
        return layernorm_fwd(xs[0], gammas[0], betas[0], ys[0], means[0], rstds[0], rows, cols, eps)
    _lib.call("missm_layernorm_fwd_grouped", len(xs), _ptrs(xs), _ptrs(gammas), _ptrs(betas), _ptrs(ys), _ptrs(means), _ptrs(rstds), rows, cols,
              float(eps), dt(ys[0]), _s())


def layernorm_bwd_lanes(dys, xs, means, rstds, gammas, dxs, dgammas, dbetas, rows, cols, dx_casts):
    """residual-stream backward (accumulate into dx, dy-dtype copy to dx_cast) of several lock-step towers in one launch"""
    if len(dys) == 1:
        return layernorm_bwd(dys[0], xs[0], means[0], rstds[0], gammas[0], dxs[0], dgammas[0], dbetas[0], rows, cols, accumulate=True,
                             dx_cast=dx_casts[0])
    if any(c.dtype != dys[0].dtype for c in dx_casts):
        raise _lib.MissmError("layernorm_bwd_lanes: dx_cast must have dy's dtype")
    _lib.call("missm_layernorm_bwd_grouped", len(dys), _ptrs(dys), _ptrs(xs), _ptrs(means), _ptrs(rstds), _ptrs(gammas), _ptrs(dxs), _ptrs(dgammas),
              _ptrs(dbetas), _ptrs(dx_casts), rows, cols, dt(dys[0]), _s())


def cast_rows(x, out, R, C, rdiv=0, roff=0):
    _lib.call("missm_cast_rows", x.data_ptr(), out.data_ptr(), R, C, rdiv, roff, dt(out), _s())
    return out


def mean_rows(x, out, B, T, cols):
    _lib.call("missm_mean_rows", x.data_ptr(), out.data_ptr(), B, T, cols, _s())
    return out


def attention_fwd(qkv, out, lse, nseq, L, H, hd, *, seq_div=1, seq_outer=None, seq_inner=0, tok_stride=1, causal=False,
                  key_mask=None, scale=None):
    seq_outer = L if seq_outer is None else seq_outer
    scale = hd ** -0.5 if scale is None else scale
    rows_needed = ((nseq - 1) // seq_div) * seq_outer + ((nseq - 1) % seq_div) * seq_inner + (L - 1) * tok_stride + 1
    if qkv.shape[0] < rows_needed or out.shape[0] < rows_needed or qkv.shape[1] < 3 * H * hd or out.shape[1] < H * hd:
        raise _lib.MissmError("attention_fwd: row addressing exceeds the buffers")
    if key_mask is not None and (key_mask.dtype != torch.int32 or key_mask.numel() < nseq * L):
        raise _lib.MissmError("attention_fwd: key_mask must be int32 [nseq, L]")
    if lse is not None and lse.numel() < nseq * H * L:
        raise _lib.MissmError("attention_fwd: lse too small")
    prof = ATTN_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("missm_attention_fwd", qkv.data_ptr(), out.data_ptr(), _p(lse), nseq, L, H, hd, qkv.stride(0), out.stride(0),
              seq_div, seq_outer, seq_inner, tok_stride, int(causal), _p(key_mask), float(scale), dt(qkv), _s())
    if prof is not None:
        e1.record()
        es = qkv.element_size()
        prof.append((e0, e1, 4.0 * nseq * H * L * L * hd, nseq * L * H * hd * 4 * es, "fwd"))      # read q,k,v + write out
    return out


def attention_bwd(qkv, out, dout, lse, dqkv, nseq, L, H, hd, *, seq_div=1, seq_outer=None, seq_inner=0, tok_stride=1, causal=False,
                  key_mask=None, scale=None):
    seq_outer = L if seq_outer is None else seq_outer
    scale = hd ** -0.5 if scale is None else scale
    rows_needed = ((nseq - 1) // seq_div) * seq_outer + ((nseq - 1) % seq_div) * seq_inner + (L - 1) * tok_stride + 1
    if min(qkv.shape[0], dout.shape[0], out.shape[0], dqkv.shape[0]) < rows_needed or dqkv.stride(0) != qkv.stride(0) or \
            out.stride(0) != dout.stride(0) or out.dtype != qkv.dtype:
        raise _lib.MissmError("attention_bwd: row addressing exceeds the buffers")
    prof = ATTN_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("missm_attention_bwd", qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), nseq, L, H, hd,
              qkv.stride(0), dout.stride(0), seq_div, seq_outer, seq_inner, tok_stride, int(causal), _p(key_mask), float(scale),
              dt(qkv), _s())
    if prof is not None:
        e1.record()
        es = qkv.element_size()
        prof.append((e0, e1, 10.0 * nseq * H * L * L * hd, nseq * L * H * hd * 8 * es, "bwd"))     # read q,k,v,o,do + write dq,dk,dv
    return dqkv


def unfold_patches(pixels: torch.Tensor, out: torch.Tensor, ps: int):
    """pixels fp32 [N,C,H,W] or [B,C,T,H,W] -> out[(n p), Kp], Kp = C*ps*ps rounded up to a multiple of 8 (zero columns behind K)."""
    if pixels.dim() == 4:
        B, Cc, H, W = pixels.shape
        T, sb, st, sc = 1, pixels.stride(0), 0, pixels.stride(1)
    else:
        B, Cc, T, H, W = pixels.shape
        sb, sc, st = pixels.stride(0), pixels.stride(1), pixels.stride(2)
    if pixels.stride(-1) != 1 or pixels.stride(-2) != W or pixels.dtype != torch.float32:
        raise _lib.MissmError("unfold_patches: need fp32 pixels with contiguous HxW planes")
    P = (H // ps) * (W // ps)
    if out.shape[0] < B * T * P or out.shape[1] != (Cc * ps * ps + 7) // 8 * 8 or not out.is_contiguous():
        raise _lib.MissmError("unfold_patches: bad output")
    _lib.call("missm_unfold_patches", pixels.data_ptr(), out.data_ptr(), B, T, Cc, H, W, ps, sb, st, sc, dt(out), _s())
    return out


def embed_assemble(patches, cls, pos, x, N, S, d):
    _lib.call("missm_embed_assemble", patches.data_ptr(), cls.data_ptr(), pos.data_ptr(), x.data_ptr(), N, S, d, dt(patches), _s())
    return x


def _i64(t, what: str):
    """index tensors cross the C ABI as ``const long*``: anything else would be reinterpreted, not converted"""
    if t is not None and (t.dtype != torch.int64 or not t.is_contiguous()):
        raise _lib.MissmError(f"{what} must be a contiguous int64 tensor (got {t.dtype})")
    return t


def token_embed_fwd(ids, tok, pos, h, B, S, d):
    _i64(ids, "token_embed_fwd: input_ids")
    _lib.call("missm_token_embed_fwd", ids.data_ptr(), tok.data_ptr(), pos.data_ptr(), h.data_ptr(), B, S, d, tok.shape[0], _s())
    return h


def token_embed_bwd(ids, dh, dtok, dpos, B, S, d):
    _i64(ids, "token_embed_bwd: input_ids")
    _lib.call("missm_token_embed_bwd", ids.data_ptr(), dh.data_ptr(), dtok.data_ptr(), dpos.data_ptr(), B, S, d, dtok.shape[0], _s())


def argmax_rows(ids, out, B, S):
    _i64(ids, "argmax_rows: input_ids")
    _lib.call("missm_argmax_rows", ids.data_ptr(), out.data_ptr(), B, S, _s())
    return out


def masked_copy_block(dst, src, row_code, code, keep_matching: bool = False):
    """dst = where(row_code == code, 0, src) for two [B, W] views with unit inner stride (keep_matching: where(..., src, 0))"""
    B, W = src.shape
    if dst.shape != src.shape or dst.stride(1) != 1 or src.stride(1) != 1:
        raise _lib.MissmError("masked_copy_block: bad layout")
    _i64(row_code, "masked_copy_block: row codes")
    _lib.call("missm_masked_copy_block", dst.data_ptr(), dst.stride(0), src.data_ptr(), src.stride(0), B, W, _p(row_code), int(code),
              int(keep_matching), _s())


def add_block(dst, src):
    """dst += src for two [B, W] views with unit inner stride (column slices of wider buffers)"""
    B, W = src.shape
    if dst.shape != src.shape or dst.stride(1) != 1 or src.stride(1) != 1:
        raise _lib.MissmError("add_block: bad layout")
    _lib.call("missm_add_block", dst.data_ptr(), dst.stride(0), src.data_ptr(), src.stride(0), B, W, _s())


def small_linear_fwd(x, w, bias, y, *, relu=False, row_code=None, code=0, x_sub=None, select=False, alpha=1.0, accumulate=False):
    """y may be a column slice of a wider row-major buffer (its row stride is passed on)."""
    B, I = x.shape
    O = w.shape[0]
    if y.shape[0] < B or y.shape[1] != O or y.stride(1) != 1 or not x.is_contiguous() or not w.is_contiguous():
        raise _lib.MissmError("small_linear_fwd: bad operand layout")
    _i64(row_code, "small_linear_fwd: row codes (missing_index)")
    _lib.call("missm_small_linear_fwd", x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), B, I, O, y.stride(0), int(relu),
              _p(row_code), int(code), _p(x_sub), int(select), float(alpha), int(accumulate), _s())
    return y


def gate_fwd(d, pre, y, *, row_code=None, code=0, accumulate=False):
    B, F = pre.shape
    _i64(row_code, "gate_fwd: row codes (missing_index)")
    _lib.call("missm_gate_fwd", d.data_ptr(), d.stride(0), pre.data_ptr(), y.data_ptr(), B, F, _p(row_code), int(code), int(accumulate), _s())
    return y


def gate_bwd(dy, d, pre, dd, dpre, *, row_code=None, code=0, accumulate_dd=False):
    B, F = pre.shape
    _i64(row_code, "gate_bwd: row codes (missing_index)")
    _lib.call("missm_gate_bwd", dy.data_ptr(), d.data_ptr(), d.stride(0), pre.data_ptr(), dd.data_ptr(), dd.stride(0), dpre.data_ptr(), B, F,
              _p(row_code), int(code), int(accumulate_dd), _s())


def small_linear_bwd(dy, x, w, dx, dw, dbias, *, relu_y=None, row_code=None, code=0, x_sub=None, select=False, alpha=1.0,
                     accumulate_dx=False, accumulate_dw=False):
    B, I = x.shape
    O = w.shape[0]
    if dy.shape[1] != O or dy.stride(1) != 1:
        raise _lib.MissmError("small_linear_bwd: bad dy layout")
    _i64(row_code, "small_linear_bwd: row codes (missing_index)")
    _lib.call("missm_small_linear_bwd", dy.data_ptr(), dy.stride(0), x.data_ptr(), w.data_ptr(), _p(relu_y), _p(dx), _p(dw), _p(dbias),
              B, I, O, _p(row_code), int(code), _p(x_sub), int(select), float(alpha), int(accumulate_dx), int(accumulate_dw), _s())


def l2norm_scale_fwd(x, y, scale):
    _lib.call("missm_l2norm_scale_fwd", x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], float(scale), _s())
    return y


def l2norm_scale_bwd(dy, x, dx, scale):
    _lib.call("missm_l2norm_scale_bwd", dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.shape[0], x.shape[1], float(scale), _s())
    return dx


def cross_entropy(logits, labels, loss, dlogits):
    _i64(labels, "cross_entropy: labels")
    if labels.numel() != logits.shape[0]:
        raise _lib.MissmError("cross_entropy: one label per row of logits")
    _lib.call("missm_cross_entropy", logits.data_ptr(), labels.data_ptr(), loss.data_ptr(), _p(dlogits), logits.shape[0],
              logits.shape[1], _s())
    return loss


def kl_loss(student, teacher, loss, dstudent, temperature, row_mask=None):
    B, Cn = student.shape
    if teacher.shape != student.shape or not student.is_contiguous() or not teacher.is_contiguous() or student.dtype != torch.float32:
        raise _lib.MissmError("kl_loss: student / teacher must be contiguous fp32 [B, C] of one shape")
    if row_mask is not None and (row_mask.dtype not in (torch.bool, torch.uint8) or row_mask.numel() != B or not row_mask.is_contiguous()):
        raise _lib.MissmError("kl_loss: row_mask must be a contiguous bool / uint8 [B]")
    _lib.call("missm_kl_loss", student.data_ptr(), teacher.data_ptr(), _p(row_mask), loss.data_ptr(), _p(dstudent), B, Cn,
              float(temperature), _s())
    return loss


def mse_loss(a, b, loss, da):
    if a.shape != b.shape or not a.is_contiguous() or not b.is_contiguous() or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise _lib.MissmError("mse_loss: operands must be contiguous fp32 of one shape")
    _lib.call("missm_mse_loss", a.data_ptr(), b.data_ptr(), loss.data_ptr(), _p(da), a.numel(), _s())
    return loss


def ema_update(teacher, student, decay):
    if teacher.shape != student.shape or not teacher.is_contiguous() or not student.is_contiguous() or teacher.dtype != torch.float32:
        raise _lib.MissmError("ema_update: operands must be contiguous fp32 of one shape")
    _lib.call("missm_ema_update", teacher.data_ptr(), student.data_ptr(), teacher.numel(), float(decay), _s())


def preprocess_image(src, dst, *, chw: bool, pre_scale: float, pre_min: float, pre_max: float, pre_div: float, mean, std):
    """one decoded image (device uint8 / fp32, [C,H,W] or [H,W,C], C = 1 or 3) -> dst fp32 [3, S, S]: resize of the shorter edge to S
    (antialiased bicubic), centre crop, normalisation - see include/missm_hip.h"""
    import ctypes as C
    _req(src, "preprocess_image src"); _req(dst, "preprocess_image dst")
    if src.dtype not in (torch.uint8, torch.float32) or not src.is_contiguous() or src.dim() != 3:
        raise _lib.MissmError("preprocess_image: src must be a contiguous uint8 / float32 image [C,H,W] or [H,W,C]")
    Cc, H, W = (src.shape if chw else (src.shape[2], src.shape[0], src.shape[1]))
    if dst.dtype != torch.float32 or not dst.is_contiguous() or dst.dim() != 3 or dst.shape[0] != 3 or dst.shape[1] != dst.shape[2]:
        raise _lib.MissmError("preprocess_image: dst must be a contiguous fp32 [3, S, S]")
    m3, s3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
    _lib.call("missm_preprocess_image", src.data_ptr(), int(src.dtype == torch.uint8), int(chw), int(H), int(W), int(Cc), dst.data_ptr(),
              int(dst.shape[1]), float(pre_scale), float(pre_min), float(pre_max), float(pre_div), m3, s3, _s())
    return dst


def gelu_bwd(dy, pre, dx):
    if not (dy.is_contiguous() and pre.is_contiguous() and dx.is_contiguous()) or dy.dtype != torch.float32 or dy.numel() != pre.numel():
        raise _lib.MissmError("gelu_bwd: fp32 contiguous operands of one size")
    _lib.call("missm_gelu_bwd", dy.data_ptr(), pre.data_ptr(), dx.data_ptr(), dy.numel(), _s())
    return dx


def sgat_fwd(xp, att_l, att_r, node_ok, out, alpha, B, M, H, Cc, bias=None, out_gelu=None):
    for t in (xp, att_l, att_r, out, alpha):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise _lib.MissmError("sgat_fwd: fp32 contiguous operands")
    if node_ok.dtype not in (torch.bool, torch.uint8) or node_ok.numel() != B * M or not node_ok.is_contiguous():
        raise _lib.MissmError("sgat_fwd: node_ok must be a contiguous bool [B, M]")
    if xp.numel() != B * M * H * Cc or out.numel() != xp.numel() or alpha.numel() != B * H * M * M or att_l.numel() != H * Cc or att_r.numel() != H * Cc:
        raise _lib.MissmError("sgat_fwd: operand sizes do not match B, M, H, C")
    if bias is not None and (bias.numel() != H * Cc or bias.dtype != torch.float32 or not bias.is_contiguous()):
        raise _lib.MissmError("sgat_fwd: bias must be fp32 [H * C]")
    _lib.call("missm_sgat_fwd", xp.data_ptr(), att_l.data_ptr(), att_r.data_ptr(), node_ok.data_ptr(), _p(bias), out.data_ptr(), _p(out_gelu),
              alpha.data_ptr(), B, M, H, Cc, _s())


def sgat_bwd(xp, att_l, att_r, node_ok, alpha, dout, dxp, dl_part, dr_part, B, M, H, Cc):
    for t in (xp, att_l, att_r, alpha, dout, dxp, dl_part, dr_part):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise _lib.MissmError("sgat_bwd: fp32 contiguous operands")
    if dout.numel() != B * M * H * Cc or dxp.numel() != dout.numel() or dl_part.numel() != B * H * Cc or dr_part.numel() != B * H * Cc:
        raise _lib.MissmError("sgat_bwd: operand sizes do not match B, M, H, C")
    _lib.call("missm_sgat_bwd", xp.data_ptr(), att_l.data_ptr(), att_r.data_ptr(), node_ok.data_ptr(), alpha.data_ptr(), dout.data_ptr(),
              dxp.data_ptr(), dl_part.data_ptr(), dr_part.data_ptr(), B, M, H, Cc, _s())


def dropout_fwd(x, y, mask, p, seed):
    _lib.call("missm_dropout_fwd", x.data_ptr(), y.data_ptr(), mask.data_ptr(), x.numel(), float(p), int(seed), _s())
    return y


def dropout_bwd(dy, mask, dx, p):
    _lib.call("missm_dropout_bwd", dy.data_ptr(), mask.data_ptr(), dx.data_ptr(), dy.numel(), float(p), _s())
    return dx


def adam_cast_batched(table, ntiles, master, g, m, v, step, lr, beta1, beta2, eps, weight_decay, grad_scale, dtype_code, tile_start=0):
    """Adam on the weight matrices of a cast-tile table + refresh of their compute-dtype copies; g, m, v are flat buffers
    parallel to ``master`` (same element offsets).  ``tile_start`` / ``ntiles`` select a sub-range of the table (one gradient
    bucket's matrices)."""
    if not (g.numel() == m.numel() == v.numel() == master.numel()) or g.dtype != torch.float32:
        raise _lib.MissmError("adam_cast_batched: g, m, v must be fp32 buffers parallel to the master buffer")
    off = lambda t: (t.data_ptr() - master.data_ptr()) // 4   # noqa: E731
    if ntiles <= 0:
        return
    if tile_start < 0 or (tile_start + ntiles) * 40 > table.numel():
        raise _lib.MissmError("adam_cast_batched: tile range outside the table")
    _lib.call("missm_adam_cast_batched", table.data_ptr() + 40 * tile_start, ntiles, off(g), off(m), off(v), int(step), float(lr), float(beta1),
              float(beta2), float(eps), float(weight_decay), float(grad_scale), dtype_code, _s())


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    _lib.call("missm_adam_step", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), int(step), float(lr),
              float(beta1), float(beta2), float(eps), float(weight_decay), float(grad_scale), _s())


def lora_merge(w, a, b, scale):
    """w[n_out, k_in] (fp32 view, unit inner stride) += scale * b[n_out, r] @ a[r, k_in]  (peft's merged weight, see include/missm_hip.h)"""
    n, k = w.shape
    r = a.shape[0]
    if a.shape != (r, k) or b.shape != (n, r) or w.stride(1) != 1 or not a.is_contiguous() or not b.is_contiguous() or w.dtype != torch.float32:
        raise _lib.MissmError("lora_merge: w [n, k], a [r, k], b [n, r] (fp32, contiguous rows)")
    _lib.call("missm_lora_merge", w.data_ptr(), w.stride(0), a.data_ptr(), b.data_ptr(), n, k, r, float(scale), _s())


def lora_grad(g, a, b, da, db, scale):
    """da[r, k_in] += scale * b^T g ; db[n_out, r] += scale * g a^T  from the full weight gradient g[n_out, k_in]"""
    n, k = g.shape
    r = a.shape[0]
    if a.shape != (r, k) or b.shape != (n, r) or da.shape != a.shape or db.shape != b.shape or g.stride(1) != 1 or g.dtype != torch.float32 or \
            not (a.is_contiguous() and b.is_contiguous() and da.is_contiguous() and db.is_contiguous()):
        raise _lib.MissmError("lora_grad: g [n, k], a / da [r, k], b / db [n, r] (fp32, contiguous rows)")
    _lib.call("missm_lora_grad", g.data_ptr(), g.stride(0), a.data_ptr(), b.data_ptr(), da.data_ptr(), db.data_ptr(), n, k, r, float(scale), _s())


def kaldi_fbank(wave, num_mel_bins, sample_rate, *, frame_length=25.0, frame_shift=10.0, low_freq=20.0, high_freq=0.0, preemphasis=0.97,
                subtract_global_mean=True):
    """wave: device fp32 [n] (one channel) -> log mel filter-bank energies [frames, num_mel_bins] (see include/missm_hip.h)"""
    _req(wave, "kaldi_fbank wave")
    if wave.dim() != 1 or wave.dtype != torch.float32 or not wave.is_contiguous():
        raise _lib.MissmError("kaldi_fbank: wave must be a contiguous fp32 [n] tensor")
    n = wave.numel()
    frames = _lib.load().missm_fbank_frames(n, float(sample_rate), float(frame_length), float(frame_shift))
    if frames <= 0:
        raise _lib.MissmError("kaldi_fbank: waveform shorter than one frame")
    gm = None
    if subtract_global_mean:
        gm = torch.empty(1, device=wave.device, dtype=torch.float32)
        _lib.call("missm_buffer_mean", wave.data_ptr(), n, gm.data_ptr(), _s())
    out = torch.empty(frames, num_mel_bins, device=wave.device, dtype=torch.float32)
    _lib.call("missm_kaldi_fbank", wave.data_ptr(), n, _p(gm), out.data_ptr(), int(num_mel_bins), float(sample_rate), float(frame_length),
              float(frame_shift), float(low_freq), float(high_freq), float(preemphasis), _s())
    return out


def mel_assemble(mel, target_length, starts, mean, std):
    frames, nb = mel.shape
    out = torch.empty(3, nb, target_length, device=mel.device, dtype=torch.float32)
    _lib.call("missm_mel_assemble", mel.data_ptr(), frames, nb, out.data_ptr(), int(target_length), int(starts[0]), int(starts[1]), int(starts[2]),
              float(mean), float(std), _s())
    return out


def sinc_resample(wave, kernels, orig_freq, new_freq, width, n_out):
    """wave fp32 [n] -> fp32 [n_out]; kernels fp32 [new_freq, 2 * width + orig_freq] (frequencies already divided by their gcd)"""
    _req(wave, "sinc_resample wave"); _req(kernels, "sinc_resample kernels")
    if wave.dim() != 1 or kernels.dim() != 2 or kernels.shape[0] != new_freq or not kernels.is_contiguous() or wave.dtype != torch.float32:
        raise _lib.MissmError("sinc_resample: wave [n] and kernels [new_freq, 2 width + orig_freq], fp32")
    out = torch.empty(n_out, device=wave.device, dtype=torch.float32)
    _lib.call("missm_sinc_resample", wave.data_ptr(), wave.numel(), kernels.data_ptr(), kernels.shape[1], int(orig_freq), int(new_freq), int(width),
              out.data_ptr(), int(n_out), _s())
    return out
