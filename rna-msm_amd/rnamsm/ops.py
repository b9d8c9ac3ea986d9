"""Per-kernel Python entry points over the C ABI (torch tensors in, raw device pointers out).

PyTorch is plumbing here: it allocates HBM and owns the HIP stream; every FLOP runs in
librnamsm_hip.so.  All functions require contiguous float32 tensors on a HIP device and raise
otherwise -- there is no eager fallback.
"""
from __future__ import annotations

import functools
import math
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ACT_GELU_ERF, ACT_NONE, F32

HEAD_DIM = 64


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _first_tensor(args):
    for a in args:
        if isinstance(a, torch.Tensor):
            return a
        if isinstance(a, (tuple, list)):
            t = _first_tensor(a)
            if t is not None:
                return t
    return None


def _on_operand_device(fn):
    """The library launches on the calling thread's CURRENT device and stream.  A worker thread (the CLI's MSA reader)
    or a model on cuda:1 has operands elsewhere, so every op enters the device of its first tensor operand first;
    `_stream()` then is that device's current stream."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        t = _first_tensor(args)
        if t is None:
            t = _first_tensor(tuple(kwargs.values()))
        if t is None or not t.is_cuda:
            return fn(*args, **kwargs)            # the op itself raises the "no CPU path" error
        with torch.cuda.device(t.device):
            return fn(*args, **kwargs)
    return wrapper


def _dev(t: torch.Tensor, name: str, dtype=torch.float32) -> int:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.RnamsmError(f"{name}: expected a tensor on the HIP device (no CPU path exists)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t.data_ptr()


def _rowmajor(t: torch.Tensor, name: str) -> int:
    """Leading dimension (in elements) of a 2-D view whose last dim is contiguous."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: expected a 2-D tensor with contiguous rows, got strides {t.stride()}")
    return t.stride(0)


def set_param(name: str, value: int) -> None:
    """rnamsm_set_param: in-process A/B knobs (include/rnamsm.h lists them)."""
    _lib.check(_lib.load().rnamsm_set_param(name.encode(), int(value)))


def get_param(name: str) -> int:
    return int(_lib.load().rnamsm_get_param(name.encode()))


@_on_operand_device
def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    D = x.shape[-1]
    x2 = x.contiguous().view(-1, D)
    y = torch.empty_like(x2) if out is None else out
    _lib.check(_lib.load().rnamsm_layernorm(_dev(x2, "x"), _dev(gamma, "gamma"), _dev(beta, "beta"), _dev(y, "out"),
                                            x2.shape[0], D, eps, _stream()))
    return y.view(x.shape)


@_on_operand_device
def layernorm_split(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, split: int = 3, fmt: int = 1):
    """LayerNorm whose output goes straight into 16-bit planes (hi, lo | None) [T, D]: the A operand of a 16-bit GEMM."""
    D = x.shape[-1]
    x2 = x.contiguous().view(-1, D)
    hi = torch.empty(x2.shape, dtype=torch.int16, device=x2.device)
    lo = torch.empty(x2.shape, dtype=torch.int16, device=x2.device) if split == 3 else None
    _lib.check(_lib.load().rnamsm_layernorm_split(_dev(x2, "x"), _dev(gamma, "gamma"), _dev(beta, "beta"), hi.data_ptr(),
                                                  None if lo is None else lo.data_ptr(), x2.shape[0], D, eps, fmt, _stream()))
    return hi, lo


@_on_operand_device
def linear(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
           residual: Optional[torch.Tensor] = None, scale: float = 1.0, scale_cols: int = 0,
           out: Optional[torch.Tensor] = None, zero_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act((a @ w.T + bias) * (col < scale_cols ? scale : 1)) + residual;  a [M,K], w [N,K].
    zero_rows (uint8 [M]): flagged rows get 0 in the scaled columns (q *= 1 - padding_mask)."""
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    _lib.check(_lib.load().rnamsm_gemm_bias_act_res(
        _dev(a, "a"), _rowmajor(a, "a"), _dev(w.contiguous(), "w"), None if bias is None else _dev(bias, "bias"),
        None if residual is None else _dev(residual, "residual"), 0 if residual is None else _rowmajor(residual, "residual"),
        _dev(out, "out"), _rowmajor(out, "out"), M, N, K, act, scale, scale_cols,
        None if zero_rows is None else _dev(zero_rows, "zero_rows", torch.uint8), F32, _stream()))
    return out


@_on_operand_device
def linear_row_scaled(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], row_factor: torch.Tensor,
                      scale: float = 1.0, scale_cols: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[m, n] = ((a @ w.T + bias)[m, n] * scale) * row_factor[m] for n < scale_cols, (a @ w.T + bias)[m, n] elsewhere
    (rnamsm_gemm_row_scaled: the QKV projection of a ragged batch; the general form of linear(zero_rows=...))."""
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    _lib.check(_lib.load().rnamsm_gemm_row_scaled(
        _dev(a, "a"), _rowmajor(a, "a"), _dev(w.contiguous(), "w"), None if bias is None else _dev(bias, "bias"),
        _dev(out, "out"), _rowmajor(out, "out"), M, N, K, scale, scale_cols, _dev(row_factor, "row_factor"), F32, _stream()))
    return out


@_on_operand_device
def ln_fold_weights(w: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """LayerNorm(gamma, beta) folded into the Linear (w [N,K], bias) that consumes it: (Wg [N,K], c [N], d [N]) with
    LN(x) w^T + bias = rstd * (x Wg^T - mean * c) + d  (include/rnamsm.h, K1 folded)."""
    w = w.contiguous()
    N, K = w.shape
    wg = torch.empty_like(w)
    c = torch.empty(N, device=w.device, dtype=torch.float32)
    d = torch.empty(N, device=w.device, dtype=torch.float32)
    _lib.check(_lib.load().rnamsm_ln_fold_weights(_dev(w, "w"), None if bias is None else _dev(bias, "bias"),
                                                  _dev(gamma, "gamma"), _dev(beta, "beta"), wg.data_ptr(), c.data_ptr(),
                                                  d.data_ptr(), N, K, _stream()))
    return wg, c, d


@_on_operand_device
def row_partials(x: torch.Tensor) -> torch.Tensor:
    """(sum x, sum (x - slab mean)^2) of every row of x [T, D] per 32-feature slab, slab-major: [D/32, T, 2], the statistics
    format of the folded LayerNorm (include/rnamsm.h, K1 folded)."""
    D = x.shape[-1]
    x2 = x.contiguous().view(-1, D)
    out = torch.empty(D // 32, x2.shape[0], 2, device=x2.device, dtype=torch.float32)
    _lib.check(_lib.load().rnamsm_row_partials(_dev(x2, "x"), out.data_ptr(), x2.shape[0], D, _stream()))
    return out


@_on_operand_device
def linear_residual_stats(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], residual: torch.Tensor,
                          out: Optional[torch.Tensor] = None):
    """(out, row_partials): out = a @ w.T + bias + residual (may be in place: out = residual) and the row partial sums
    [N/32, M, 2] of the stored out -- the producer side of the folded LayerNorm."""
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    part = torch.empty(N // 32, M, 2, device=a.device, dtype=torch.float32)
    _lib.check(_lib.load().rnamsm_gemm_residual_stats(
        _dev(a, "a"), _rowmajor(a, "a"), _dev(w.contiguous(), "w"), None if bias is None else _dev(bias, "bias"),
        _dev(residual, "residual"), _rowmajor(residual, "residual"), _dev(out, "out"), _rowmajor(out, "out"), M, N, K,
        part.data_ptr(), M, F32, _stream()))
    return out, part


@_on_operand_device
def row_stats_from_partials(partials: torch.Tensor, K: int, rows: Optional[int] = None, eps: float = 1e-5,
                            cond_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(mean, rstd) [rows, 2] of the first `rows` rows from their slab partials [K/32, M, 2] (Chan et al.'s combination).
    cond_flag (int32 [1]): bit 1 is set when a row's mean^2 exceeds 1024 (var + eps) -- the folded GEMM then loses more than
    5 bits to cancellation and the caller should use layernorm + linear for that input."""
    M = partials.shape[1]
    rows = M if rows is None else rows
    stats = torch.empty(rows, 2, device=partials.device, dtype=torch.float32)
    _lib.check(_lib.load().rnamsm_row_stats_from_partials(_dev(partials, "partials"), M, rows, K, eps, stats.data_ptr(),
                                                          None if cond_flag is None else _dev(cond_flag, "cond_flag", torch.int32),
                                                          _stream()))
    return stats


@_on_operand_device
def linear_lnfold(x: torch.Tensor, wg: torch.Tensor, c: torch.Tensor, d: torch.Tensor, stats: Optional[torch.Tensor] = None,
                  eps: float = 1e-5, act: int = ACT_NONE, scale: float = 1.0, scale_cols: int = 0,
                  out: Optional[torch.Tensor] = None, cond_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act((rstd * (x @ wg.T - mean * c) + d) * (col < scale_cols ? scale : 1)) = act(Linear(LayerNorm(x)) ...): the GEMM
    reads x itself; (wg, c, d) from ln_fold_weights; (mean, rstd) of row m = stats[m] (row_stats_from_partials; stats may
    cover more rows than x), or summed by the GEMM itself when None (which then also reports the precondition in cond_flag)."""
    M, K = x.shape
    N = wg.shape[0]
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().rnamsm_gemm_lnfold(_dev(x, "x"), _rowmajor(x, "x"), _dev(wg, "wg"), _dev(c, "c"), _dev(d, "d"),
                                              eps, None if stats is None else _dev(stats, "stats"),
                                              None if cond_flag is None else _dev(cond_flag, "cond_flag", torch.int32),
                                              _dev(out, "out"), _rowmajor(out, "out"), M, N, K, act, scale, scale_cols, F32,
                                              _stream()))
    return out


@_on_operand_device
def split_bf16(w: torch.Tensor, want_lo: bool = True, fmt: int = 0):
    """fp32 tensor -> (hi, lo) 16-bit planes (fmt 0 = bf16, 1 = fp16) as int16 tensors of the same shape
    (lo = half(w - hi), None if not wanted)."""
    w = w.contiguous()
    hi = torch.empty(w.shape, dtype=torch.int16, device=w.device)
    lo = torch.empty(w.shape, dtype=torch.int16, device=w.device) if want_lo else None
    _lib.check(_lib.load().rnamsm_split_bf16(_dev(w, "w"), hi.data_ptr(), None if lo is None else lo.data_ptr(),
                                             w.numel(), fmt, _stream()))
    return hi, lo


@_on_operand_device
def linear_bf16(a: torch.Tensor, w_hi: torch.Tensor, w_lo: Optional[torch.Tensor], bias: Optional[torch.Tensor] = None,
                act: int = ACT_NONE, residual: Optional[torch.Tensor] = None, scale: float = 1.0, scale_cols: int = 0,
                out: Optional[torch.Tensor] = None, split: int = 3, fmt: int = 1) -> torch.Tensor:
    """`linear` on the 16-bit matrix cores: split = 1 (bf16 operands) or 3 (hi/lo fp16 split: fmt must be 1 = f16x3)."""
    M, K = a.shape
    N = w_hi.shape[0]
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    _lib.check(_lib.load().rnamsm_gemm_bf16(
        _dev(a, "a"), _rowmajor(a, "a"), _dev(w_hi, "w_hi", torch.int16), None if w_lo is None else _dev(w_lo, "w_lo", torch.int16),
        None if bias is None else _dev(bias, "bias"), None if residual is None else _dev(residual, "residual"),
        0 if residual is None else _rowmajor(residual, "residual"), _dev(out, "out"), _rowmajor(out, "out"), M, N, K, act,
        scale, scale_cols, split, fmt, None, None, None, None, _stream()))
    return out


@_on_operand_device
def row_logits(q: torch.Tensor, k: torch.Tensor, R: int, C: int, H: int, rows_per_chunk: int = 0) -> Tuple[torch.Tensor, int]:
    """q, k: [R*C, *] views with row stride ld; returns (partial [nsplit,H,C,C], nsplit).  rows_per_chunk > 0: one slab
    per reference row chunk (the padded, chunked path: rnamsm_row_logits_chunked)."""
    lib = _lib.load()
    nsplit = (R + rows_per_chunk - 1) // rows_per_chunk if rows_per_chunk > 0 else lib.rnamsm_row_logits_nsplit(R, C, H)
    partial = torch.empty(nsplit, H, C, C, device=q.device, dtype=torch.float32)
    ld = _rowmajor(q, "q")
    assert _rowmajor(k, "k") == ld
    if rows_per_chunk > 0:
        _lib.check(lib.rnamsm_row_logits_chunked(_dev(q, "q"), _dev(k, "k"), ld, _dev(partial, "partial"), R, C, H, HEAD_DIM,
                                                 rows_per_chunk, F32, _stream()))
    else:
        _lib.check(lib.rnamsm_row_logits(_dev(q, "q"), _dev(k, "k"), ld, _dev(partial, "partial"), R, C, H, HEAD_DIM, F32,
                                         _stream()))
    return partial, nsplit


def row_chunks(R: int, C: int, max_tokens_per_msa: int):
    """(number of row chunks, rows per chunk) of the reference's _batched_forward (modules.py:717-750), (0, 0) when it
    takes the direct path."""
    mt = min(int(max_tokens_per_msa), 2 ** 31 - 1)
    n = _lib.load().rnamsm_row_chunks(R, C, mt)
    return (n, max(1, mt // C)) if n else (0, 0)


@_on_operand_device
def softmax_rows(partial: torch.Tensor, out: Optional[torch.Tensor] = None,
                 key_mask: Optional[torch.Tensor] = None, chunk_pad_mask: Optional[torch.Tensor] = None,
                 rows_per_chunk: int = 0, logit_scale: float = 1.0) -> torch.Tensor:
    """key_mask uint8 [C]: direct-path fill.  chunk_pad_mask uint8 [R*C] + rows_per_chunk: the chunked path's per-chunk
    fill (slab c is masked by row c*rows_per_chunk of the padding mask).  logit_scale (direct path): multiplies the summed
    logits first -- where the exact path applies 1/sqrt(R) (rnamsm_softmax_rows_scaled)."""
    nsplit, H, C, _ = partial.shape
    probs = torch.empty(H, C, C, device=partial.device, dtype=torch.float32) if out is None else out
    if chunk_pad_mask is not None:
        _lib.check(_lib.load().rnamsm_softmax_rows_chunked(_dev(partial, "partial"), nsplit, _dev(probs, "probs"), H, C,
                                                           _dev(chunk_pad_mask, "chunk_pad_mask", torch.uint8),
                                                           rows_per_chunk, _stream()))
        return probs
    _lib.check(_lib.load().rnamsm_softmax_rows_scaled(_dev(partial, "partial"), nsplit, _dev(probs, "probs"), H, C,
                                                      None if key_mask is None else _dev(key_mask, "key_mask", torch.uint8),
                                                      float(logit_scale), _stream()))
    return probs


@_on_operand_device
def row_apply(probs: torch.Tensor, v: torch.Tensor, R: int, C: int, H: int,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    ctx = torch.empty(R * C, H * HEAD_DIM, device=v.device, dtype=torch.float32) if out is None else out
    _lib.check(_lib.load().rnamsm_row_apply(_dev(probs, "probs"), _dev(v, "v"), _rowmajor(v, "v"), _dev(ctx, "ctx"),
                                            _rowmajor(ctx, "ctx"), R, C, H, HEAD_DIM, None, None, 0, F32, _stream()))
    return ctx


@_on_operand_device
def col_attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, R: int, C: int, H: int,
             out: Optional[torch.Tensor] = None, pad_mask: Optional[torch.Tensor] = None, prescaled: bool = False,
             q_rows: Optional[int] = None) -> torch.Tensor:
    """prescaled: q carries dh^-1/2 * log2(e) (rnamsm_col_attn_fused_prescaled: no running maximum in the first pass)."""
    ctx = torch.empty(R * C, H * HEAD_DIM, device=v.device, dtype=torch.float32) if out is None else out
    ld = _rowmajor(q, "q")
    assert _rowmajor(k, "k") == ld and _rowmajor(v, "v") == ld
    if prescaled:
        assert pad_mask is None
        _lib.check(_lib.load().rnamsm_col_attn_fused_prescaled(_dev(q, "q"), _dev(k, "k"), _dev(v, "v"), ld, _dev(ctx, "ctx"),
                                                               _rowmajor(ctx, "ctx"), R, C, H, HEAD_DIM, R if q_rows is None else q_rows,
                                                               _stream()))
        return ctx
    if q_rows is not None:      # only the first q_rows query rows are computed and stored (rnamsm_col_attn_fused_queries)
        _lib.check(_lib.load().rnamsm_col_attn_fused_queries(_dev(q, "q"), _dev(k, "k"), _dev(v, "v"), ld, _dev(ctx, "ctx"),
                                                             _rowmajor(ctx, "ctx"), R, C, H, HEAD_DIM, q_rows,
                                                             None if pad_mask is None else _dev(pad_mask, "pad_mask", torch.uint8),
                                                             F32, _stream()))
        return ctx
    _lib.check(_lib.load().rnamsm_col_attn_fused(_dev(q, "q"), _dev(k, "k"), _dev(v, "v"), ld, _dev(ctx, "ctx"),
                                                 _rowmajor(ctx, "ctx"), R, C, H, HEAD_DIM,
                                                 None if pad_mask is None else _dev(pad_mask, "pad_mask", torch.uint8),
                                                 None, None, 0, F32, _stream()))
    return ctx


@_on_operand_device
def col_attn_probs(q: torch.Tensor, k: torch.Tensor, R: int, C: int, H: int, pad_mask: Optional[torch.Tensor] = None,
                   scale: float = 1.0) -> torch.Tensor:
    """The probabilities the fused column attention never forms (reference modules.py:905-917), on request:
    fp32 [H, C, R, R] = softmax_j(scale * q_i . k_j) per column and head; q as the QKV GEMM left it (already scaled)."""
    probs = torch.empty(H, C, R, R, device=q.device, dtype=torch.float32)
    ld = _rowmajor(q, "q")
    assert _rowmajor(k, "k") == ld
    _lib.check(_lib.load().rnamsm_col_attn_probs(_dev(q, "q"), _dev(k, "k"), ld, _dev(probs, "probs"), R, C, H, HEAD_DIM,
                                                 None if pad_mask is None else _dev(pad_mask, "pad_mask", torch.uint8),
                                                 scale, F32, _stream()))
    return probs


# ---- 16-bit attention contractions: operands are (hi, lo) int16 plane pairs, lo = None for plain bf16 ---------------
def _pl(t: Optional[torch.Tensor], name: str):
    return None if t is None else _dev(t, name, torch.int16)


@_on_operand_device
def linear_planes(a, w, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, residual: Optional[torch.Tensor] = None,
                  scale: float = 1.0, scale_cols: int = 0, out: Optional[torch.Tensor] = None, out_planes: bool = False,
                  fmt: int = 0):
    """16-bit GEMM whose A operand is already (hi, lo | None) planes [M, K] (the LDS-DMA kernels): returns fp32 [M, N],
    or (hi, lo | None) planes when out_planes (no residual then)."""
    a_hi, a_lo = a
    w_hi, w_lo = w
    M, K = a_hi.shape
    N = w_hi.shape[0]
    split = 3 if w_lo is not None else 1
    o_hi = o_lo = None
    if out_planes:
        o_hi = torch.empty(M, N, dtype=torch.int16, device=a_hi.device)
        o_lo = torch.empty(M, N, dtype=torch.int16, device=a_hi.device) if split == 3 else None
    elif out is None:
        out = torch.empty(M, N, device=a_hi.device, dtype=torch.float32)
    _lib.check(_lib.load().rnamsm_gemm_bf16(
        None, _rowmajor(a_hi, "a_hi"), _pl(w_hi, "w_hi"), _pl(w_lo, "w_lo"), None if bias is None else _dev(bias, "bias"),
        None if residual is None else _dev(residual, "residual"), 0 if residual is None else _rowmajor(residual, "residual"),
        None if out_planes else _dev(out, "out"), N if out_planes else _rowmajor(out, "out"), M, N, K, act, scale, scale_cols,
        split, fmt, _pl(a_hi, "a_hi"), _pl(a_lo, "a_lo"), None if o_hi is None else o_hi.data_ptr(),
        None if o_lo is None else o_lo.data_ptr(), _stream()))
    return (o_hi, o_lo) if out_planes else out


@_on_operand_device
def linear_planes_lnfold(x, wg, c: torch.Tensor, d: torch.Tensor, stats: torch.Tensor, act: int = ACT_NONE, scale: float = 1.0,
                         scale_cols: int = 0, fmt: int = 0):
    """K1 folded, 16-bit modes: x = (hi, lo | None) planes of the RAW residual stream [M, K]; wg = planes of W * gamma;
    c = row sums of the wg planes' values, d = bias + W beta; stats [M, 2].  Returns the output planes (hi, lo | None)."""
    M, K = x[0].shape
    N = wg[0].shape[0]
    split = 1 if x[1] is None else 3
    ohi = torch.empty(M, N, dtype=torch.int16, device=x[0].device)
    olo = torch.empty(M, N, dtype=torch.int16, device=x[0].device) if split == 3 else None
    _lib.check(_lib.load().rnamsm_gemm16_lnfold(_pl(x[0], "x_hi"), _pl(x[1], "x_lo"), _rowmajor(x[0], "x_hi"), _pl(wg[0], "wg_hi"),
                                                _pl(wg[1], "wg_lo"), _dev(c, "c"), _dev(d, "d"), _dev(stats, "stats"),
                                                ohi.data_ptr(), None if olo is None else olo.data_ptr(), N, M, N, K, act, scale,
                                                scale_cols, split, fmt, _stream()))
    return ohi, olo


@_on_operand_device
def linear_planes_residual_stats(a, w, bias: Optional[torch.Tensor], x: torch.Tensor, fmt: int = 0):
    """x += a w^T + bias in place (fp32) on the 16-bit matrix cores; returns (x planes (hi, lo | None), row_partials
    [N/32, M, 2]) of the new x -- the producer side of the folded LayerNorm in the 16-bit modes."""
    M, K = a[0].shape
    N = w[0].shape[0]
    split = 1 if a[1] is None else 3
    xhi = torch.empty(M, N, dtype=torch.int16, device=x.device)
    xlo = torch.empty(M, N, dtype=torch.int16, device=x.device) if split == 3 else None
    part = torch.empty(N // 32, M, 2, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().rnamsm_gemm16_residual_stats(_pl(a[0], "a_hi"), _pl(a[1], "a_lo"), _rowmajor(a[0], "a_hi"),
                                                        _pl(w[0], "w_hi"), _pl(w[1], "w_lo"),
                                                        None if bias is None else _dev(bias, "bias"), _dev(x, "x"),
                                                        _rowmajor(x, "x"), M, N, K, split, fmt, xhi.data_ptr(),
                                                        None if xlo is None else xlo.data_ptr(), N, part.data_ptr(), M,
                                                        _stream()))
    return (xhi, xlo), part


@_on_operand_device
def row_logits16(q, k, R: int, C: int, H: int, fmt: int = 0, scale: float = 1.0) -> Tuple[torch.Tensor, int]:
    """q, k: (hi, lo) plane views [R*C, *] with row stride ld (halves); returns (partial [nsplit,H,C,C] fp32, nsplit);
    scale multiplies the fp32 logits (q is expected UNSCALED)."""
    lib = _lib.load()
    nsplit = lib.rnamsm_row_logits16_nsplit(R, C, H, 1 if q[1] is None else 3)
    partial = torch.empty(nsplit, H, C, C, device=q[0].device, dtype=torch.float32)
    ld = _rowmajor(q[0], "q_hi")
    _lib.check(lib.rnamsm_row_logits16(_pl(q[0], "q_hi"), _pl(q[1], "q_lo"), _pl(k[0], "k_hi"), _pl(k[1], "k_lo"), ld,
                                       _dev(partial, "partial"), R, C, H, HEAD_DIM, scale, fmt, _stream()))
    return partial, nsplit


@_on_operand_device
def softmax_rows_planes(partial: torch.Tensor, split: int = 3, fmt: int = 1, key_mask: Optional[torch.Tensor] = None,
                        plane_scale: float = 1.0):
    """softmax_rows that also returns P * plane_scale as planes (hi, lo | None) [H*C, ldp], ldp = C rounded up to 64,
    tail zeroed."""
    nsplit, H, C, _ = partial.shape
    ldp = (C + 63) // 64 * 64
    probs = torch.empty(H, C, C, device=partial.device, dtype=torch.float32)
    p_hi = torch.empty(H * C, ldp, device=partial.device, dtype=torch.int16)
    p_lo = torch.empty(H * C, ldp, device=partial.device, dtype=torch.int16) if split == 3 else None
    _lib.check(_lib.load().rnamsm_softmax_rows_planes(
        _dev(partial, "partial"), nsplit, _dev(probs, "probs"), p_hi.data_ptr(), None if p_lo is None else p_lo.data_ptr(),
        ldp, plane_scale, H, C, None if key_mask is None else _dev(key_mask, "key_mask", torch.uint8), fmt, _stream()))
    return probs, (p_hi, p_lo)


@_on_operand_device
def row_apply16(p, v, R: int, C: int, H: int, fmt: int = 0, out_scale: float = 1.0) -> torch.Tensor:
    """p: (hi, lo) planes [H*C, ldp]; v: (hi, lo) plane views [R*C, *]; returns out_scale * P v, fp32 [R*C, H*64]."""
    ctx = torch.empty(R * C, H * HEAD_DIM, device=v[0].device, dtype=torch.float32)
    _lib.check(_lib.load().rnamsm_row_apply16(_pl(p[0], "p_hi"), _pl(p[1], "p_lo"), _rowmajor(p[0], "p_hi"),
                                              _pl(v[0], "v_hi"), _pl(v[1], "v_lo"), _rowmajor(v[0], "v_hi"),
                                              _dev(ctx, "ctx"), _rowmajor(ctx, "ctx"), R, C, H, HEAD_DIM, out_scale, None, None,
                                              fmt, _stream()))
    return ctx


@_on_operand_device
def zero_plane_rows(pl, mask: torch.Tensor, ncols: int) -> None:
    """In place: zero the first ncols halves of the plane rows flagged in mask (uint8 [T]) -- q *= 1 - padding_mask."""
    _lib.check(_lib.load().rnamsm_zero_plane_rows(_pl(pl[0], "hi"), _pl(pl[1], "lo"), _dev(mask, "mask", torch.uint8),
                                                  pl[0].shape[0], ncols, _rowmajor(pl[0], "hi"), _stream()))


@_on_operand_device
def col_attn16(q, k, v, R: int, C: int, H: int, fmt: int = 0, scale: float = 1.0,
               pad_mask: Optional[torch.Tensor] = None, out_planes: bool = False, prescaled: bool = False):
    """q, k, v: (hi, lo) plane views [R*C, *] with a common row stride; returns softmax(scale * q k^T) v, fp32
    [R*C, H*64] (q UNSCALED) -- or, with out_planes, the context as the 16-bit (hi, lo | None) int16 planes the forward's
    out_proj GEMM reads (format = the operands': bf16, or fp16 for fmt 1).  prescaled (bf16 formats, no mask): the q planes
    hold q * scale * log2(e) already (rnamsm_col_attn16_prescaled: what the forward runs); `scale` is then ignored."""
    D = H * HEAD_DIM
    ld = _rowmajor(q[0], "q_hi")
    assert _rowmajor(k[0], "k_hi") == ld and _rowmajor(v[0], "v_hi") == ld
    ctx = None if out_planes else torch.empty(R * C, D, device=v[0].device, dtype=torch.float32)
    c_hi = torch.empty(R * C, D, device=v[0].device, dtype=torch.int16) if out_planes else None
    c_lo = torch.empty(R * C, D, device=v[0].device, dtype=torch.int16) if out_planes and q[1] is not None else None
    if prescaled:
        assert fmt == 0 and pad_mask is None
        _lib.check(_lib.load().rnamsm_col_attn16_prescaled(_pl(q[0], "q_hi"), _pl(q[1], "q_lo"), _pl(k[0], "k_hi"), _pl(k[1], "k_lo"),
                                                           _pl(v[0], "v_hi"), _pl(v[1], "v_lo"), ld,
                                                           None if ctx is None else _dev(ctx, "ctx"), D, R, C, H, HEAD_DIM,
                                                           _pl(c_hi, "ctx_hi"), _pl(c_lo, "ctx_lo"), _stream()))
        return (c_hi, c_lo) if out_planes else ctx
    _lib.check(_lib.load().rnamsm_col_attn16(_pl(q[0], "q_hi"), _pl(q[1], "q_lo"), _pl(k[0], "k_hi"), _pl(k[1], "k_lo"),
                                             _pl(v[0], "v_hi"), _pl(v[1], "v_lo"), ld,
                                             None if ctx is None else _dev(ctx, "ctx"), D,
                                             R, C, H, HEAD_DIM, scale,
                                             None if pad_mask is None else _dev(pad_mask, "pad_mask", torch.uint8),
                                             _pl(c_hi, "ctx_hi"), _pl(c_lo, "ctx_lo"), fmt, _stream()))
    return (c_hi, c_lo) if out_planes else ctx


@_on_operand_device
def col_attn_probs16(q, k, R: int, C: int, H: int, fmt: int = 0, scale: float = 1.0,
                     pad_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """col_attn_probs for the 16-bit modes: q, k are (hi, lo | None) plane views (q UNSCALED)."""
    probs = torch.empty(H, C, R, R, device=q[0].device, dtype=torch.float32)
    ld = _rowmajor(q[0], "q_hi")
    assert _rowmajor(k[0], "k_hi") == ld
    _lib.check(_lib.load().rnamsm_col_attn_probs16(_pl(q[0], "q_hi"), _pl(q[1], "q_lo"), _pl(k[0], "k_hi"), _pl(k[1], "k_lo"),
                                                   ld, _dev(probs, "probs"), R, C, H, HEAD_DIM,
                                                   None if pad_mask is None else _dev(pad_mask, "pad_mask", torch.uint8),
                                                   fmt, scale, _stream()))
    return probs


@_on_operand_device
def embed_ln(tokens: torch.Tensor, embed_tokens: torch.Tensor, embed_positions: torch.Tensor, row_pos: torch.Tensor,
             gamma: torch.Tensor, beta: torch.Tensor, pad_idx: int, eps: float = 1e-5, row_pos_dim: int = 1) -> torch.Tensor:
    """tokens int64 [R,C] -> x [R*C, D]; raises on token / position ids outside the tables.  row_pos: [>= R] scalars per
    alignment row (row_pos_dim 1), or flattened [>= R, D] vectors (row_pos_dim = D: the msm/ variant, msm/model.py:289-292)."""
    R, C = tokens.shape
    D = embed_tokens.shape[1]
    out = torch.empty(R * C, D, device=tokens.device, dtype=torch.float32)
    err = torch.zeros(1, device=tokens.device, dtype=torch.int32)
    _lib.check(_lib.load().rnamsm_embed_ln_rows(
        _dev(tokens.contiguous(), "tokens", torch.int64), _dev(embed_tokens, "embed_tokens"),
        _dev(embed_positions, "embed_positions"), _dev(row_pos, "row_pos"), int(row_pos_dim), _dev(gamma, "gamma"), _dev(beta, "beta"),
        _dev(out, "out"), R, C, D, embed_tokens.shape[0], embed_positions.shape[0], pad_idx, eps,
        _dev(err, "err", torch.int32), _stream()))
    if int(err.item()) != 0:
        raise IndexError("embed_ln: token or position index out of range")
    return out


@_on_operand_device
def pack_outputs(x_final: torch.Tensor, probs_all: torch.Tensor, C: int) -> Tuple[torch.Tensor, torch.Tensor]:
    NL, H = probs_all.shape[0], probs_all.shape[1]
    D = x_final.shape[-1]
    emb = torch.empty(C - 1, D, device=x_final.device, dtype=torch.float32)
    atp = torch.empty(NL * H, C - 1, C - 1, device=x_final.device, dtype=torch.float32)
    _lib.check(_lib.load().rnamsm_pack_outputs(_dev(x_final, "x_final"), _dev(probs_all, "probs_all"), _dev(emb, "emb"),
                                               _dev(atp, "atp"), C, D, NL, H, _stream()))
    return emb, atp


@_on_operand_device
def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a + b, fp32, same shape (the residual add of NormalizedResidualBlock around a foreign layer, modules.py:396).  `out` (a
    contiguous fp32 tensor of that shape) may be `a` or `b` itself: the kernel reads an element before it writes it."""
    assert a.shape == b.shape
    if out is None:
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a, dtype=torch.float32)
    else:
        assert out.shape == a.shape and out.is_contiguous() and a.is_contiguous() and b.is_contiguous()
    _lib.check(_lib.load().rnamsm_add(_dev(a, "a"), _dev(b, "b"), _dev(out, "out"), out.numel(), _stream()))
    return out


@_on_operand_device
def head_mean(probs: torch.Tensor) -> torch.Tensor:
    """probs [H, ...] fp32 -> mean over the head axis [...] (msm/multihead_attention.py:394-397)."""
    probs = probs.contiguous()
    H = probs.shape[0]
    out = torch.empty(probs.shape[1:], device=probs.device, dtype=torch.float32)
    _lib.check(_lib.load().rnamsm_head_mean(_dev(probs, "probs"), _dev(out, "out"), H, out.numel(), _stream()))
    return out


@_on_operand_device
def contact_head(row_attn: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """row_attn [NL, H, C, C] (or [NL*H, C, C]) -> contacts [C-1, C-1] (modules.py:344-366)."""
    C = row_attn.shape[-1]
    nch = row_attn.numel() // (C * C)
    lib = _lib.load()
    ws = torch.empty(lib.rnamsm_contact_head_workspace_bytes(C, nch), dtype=torch.uint8, device=row_attn.device)
    out = torch.empty(C - 1, C - 1, device=row_attn.device, dtype=torch.float32)
    _lib.check(lib.rnamsm_contact_head(_dev(row_attn.contiguous(), "row_attn"), _dev(weight.contiguous().view(-1), "weight"),
                                       _dev(bias.contiguous().view(-1), "bias"), _dev(out, "contacts"), ws.data_ptr(),
                                       ws.numel(), C, nch, _stream()))
    return out


@_on_operand_device
def greedy_select(msa_u8: torch.Tensor, num_seqs: int, mode: str = "max") -> torch.Tensor:
    """msa uint8 [N, L] on the device -> int32 [num_seqs] ascending row indices (utils/align.py:128-148)."""
    if mode not in ("max", "min"):
        raise AssertionError(mode)
    N, L = msa_u8.shape
    lib = _lib.load()
    ws = torch.empty(lib.rnamsm_greedy_select_workspace_bytes(N, L, num_seqs), dtype=torch.uint8, device=msa_u8.device)
    out = torch.empty(num_seqs, dtype=torch.int32, device=msa_u8.device)
    _lib.check(lib.rnamsm_greedy_select(_dev(msa_u8.contiguous(), "msa", torch.uint8), N, L, num_seqs,
                                        1 if mode == "min" else 0, _dev(out, "out", torch.int32), ws.data_ptr(),
                                        ws.numel(), _stream()))
    return out


@_on_operand_device
def msa_weights(msa_u8: torch.Tensor, seqid_cutoff: float = 0.2) -> torch.Tensor:
    """msa uint8 [N, L] on the device -> float64 [N] sequence weights (MSA.weights, utils/align.py:250-253); L <= 32768.
    The reference's pdist runs over the raw character bytes (utils/align.py:121-126); callers here pass TOKEN ids.  The
    two agree for the alphabet that survives from_fasta's regexes (A/C/G/U/X/-, one token per character); characters that
    fall to <unk> would be merged by the token form -- the reader raises for those before this is reached."""
    N, L = msa_u8.shape
    out = torch.empty(N, dtype=torch.float64, device=msa_u8.device)
    _lib.check(_lib.load().rnamsm_msa_weights(_dev(msa_u8.contiguous(), "msa", torch.uint8), N, L, float(seqid_cutoff),
                                              _dev(out, "weights", torch.float64), _stream()))
    return out


def depth_scaling(R: int) -> float:
    """1/sqrt(R) as the C++ drivers form it -- `1.0f / sqrtf((float)R)` in fp32, both steps correctly rounded -- so that the
    layer-wise mirror modules hand K5 the very same factor (forming it in double and rounding once can differ in the last bit)."""
    import numpy as np
    return float(np.float32(1.0) / np.sqrt(np.float32(R)))


def row_scaling(R: int) -> float:
    """RowSelfAttention.align_scaling (modules.py:713-715)."""
    return (HEAD_DIM ** -0.5) / math.sqrt(R)
