"""Host-side mirror of the reference's axial-attention modules, executing on librnamsm_hip.

Same class names, constructor arguments, forward signatures, return shapes, error behaviour and
state_dict keys as the reference (modules.py:688-945, 404-427, 369-401, 191-267 == msm/axial_attention.py,
msm/modules.py), so a checkpoint and a caller written for the reference work unchanged.  The
nn.Linear / nn.LayerNorm children only HOLD parameters (for state_dict compatibility); no torch
operator computes anything -- every forward goes through rnamsm.ops to the HIP kernels.

Scope: inference, one MSA per call (B = 1; the model loops over a batch).  Padding masks follow the
reference's direct path, or its chunked path when R*C > max_tokens_per_msa (§8 f2).  Training-mode dropout and `self_attn_mask` raise instead of silently
doing something else.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import ACT_GELU_ERF, ACT_NONE


def _check_inference(module: nn.Module, p: float) -> None:
    if module.training and p > 0.0:
        raise NotImplementedError("rnamsm implements the inference path only: call .eval() (dropout is not implemented)")


def _tokens_2d(x: torch.Tensor):
    """[R, C, B=1, D] -> ([R*C, D] view, R, C, D)."""
    if x.dim() != 4:
        raise ValueError(f"expected x of shape [R, C, B, D], got {tuple(x.shape)}")
    R, C, B, D = x.shape
    if B != 1:
        raise NotImplementedError("rnamsm processes one MSA per call (B = 1); loop over the batch dimension")
    return x.contiguous().view(R * C, D), R, C, D


class _PackedQKV:
    """Caches the fused [3D, D] q;k;v weight (one GEMM instead of three) until a parameter changes."""

    def __init__(self):
        self._key = None
        self._w = self._b = None

    def get(self, q: nn.Linear, k: nn.Linear, v: nn.Linear):
        key = tuple((p.data_ptr(), p._version) for p in (q.weight, k.weight, v.weight, q.bias, k.bias, v.bias))
        if key != self._key:
            self._w = torch.cat([q.weight.detach(), k.weight.detach(), v.weight.detach()], 0).contiguous()
            self._b = torch.cat([q.bias.detach(), k.bias.detach(), v.bias.detach()], 0).contiguous()
            self._key = key
        return self._w, self._b


GEMM_DTYPES = ("f32", "f16x3", "bf16")


def _mode_of(module: nn.Module):
    """(split, fmt) of a module's `gemm_dtype`: None for the exact-fp32 path, else (1 | 3, 0 = bf16 | 1 = fp16)."""
    mode = getattr(module, "gemm_dtype", "f32")
    if mode not in GEMM_DTYPES:
        raise ValueError(f"gemm_dtype must be one of {GEMM_DTYPES}, got {mode!r}")
    return None if mode == "f32" else (1 if mode == "bf16" else 3, 1 if mode == "f16x3" else 0)


class _WeightPlanes:
    """16-bit hi(/lo) planes of a weight tensor (rnamsm_split_bf16), rebuilt when the tensor or the format changes."""

    def __init__(self):
        self._key = None
        self._planes = None

    def get(self, weight: torch.Tensor, split: int, fmt: int):
        key = (weight.data_ptr(), weight._version, split, fmt)
        if key != self._key:
            self._planes = ops.split_bf16(weight.detach().contiguous(), want_lo=split == 3, fmt=fmt)
            self._key = key
        return self._planes


def _act_planes(x2: torch.Tensor, split: int, fmt: int):
    return ops.split_bf16(x2, want_lo=split == 3, fmt=fmt)


def _cols(planes, a: int, b: int):
    return tuple(None if p is None else p[:, a:b] for p in planes)


class _AxialAttentionBase(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0.0, max_tokens_per_msa: int = 2 ** 16):
        super().__init__()
        if embed_dim != num_heads * ops.HEAD_DIM:
            raise ValueError(f"the HIP kernels are built for head_dim 64 (embed_dim={embed_dim}, num_heads={num_heads})")
        self.num_heads = num_heads
        self.dropout = dropout
        self.head_dim = embed_dim // num_heads
        self.scaling = self.head_dim ** -0.5
        # The reference chunks rows/columns above this token budget (_batched_forward).  The kernels tile internally and
        # produce the same sums, so without padding the value changes nothing; WITH padding the chunked row path fills
        # its key mask per chunk, which RowSelfAttention reproduces.
        self.max_tokens_per_msa = max_tokens_per_msa
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self._packed = _PackedQKV()
        # arithmetic of the contractions: "f32" (exact, default) or a 16-bit matrix-core mode -- the same kernels and
        # operand formats rnamsm_forward uses for model.gemm_dtype (MSATransformer sets it on every module)
        self.gemm_dtype = "f32"
        self._wqkv_planes, self._wo_planes = _WeightPlanes(), _WeightPlanes()

    def _qkv(self, x2: torch.Tensor, q_scale: float, zero_rows=None) -> torch.Tensor:
        w, b = self._packed.get(self.q_proj, self.k_proj, self.v_proj)
        D = x2.shape[1]
        return ops.linear(x2, w, b, scale=q_scale, scale_cols=D, zero_rows=zero_rows)   # [T, 3D] = q*scale | k | v

    def _project_out(self, ctx: torch.Tensor, residual: Optional[torch.Tensor]) -> torch.Tensor:
        mode = _mode_of(self)
        if mode is not None:
            split, fmt = mode
            return ops.linear_planes(_act_planes(ctx, split, fmt), self._wo_planes.get(self.out_proj.weight, split, fmt),
                                     self.out_proj.bias.detach(), residual=residual, fmt=fmt)
        return ops.linear(ctx, self.out_proj.weight.detach(), self.out_proj.bias.detach(), residual=residual)

    def _qkv_planes(self, x2: torch.Tensor, split: int, fmt: int):
        """q | k | v as 16-bit planes [T, 3D] (q UNSCALED: the 16-bit attention kernels scale the fp32 logits)."""
        w, b = self._packed.get(self.q_proj, self.k_proj, self.v_proj)
        return ops.linear_planes(_act_planes(x2, split, fmt), self._wqkv_planes.get(w, split, fmt), b, out_planes=True, fmt=fmt)

    @staticmethod
    def _mask_bytes(self_attn_mask, self_attn_padding_mask, R: int, C: int):
        """self_attn_padding_mask [B=1, R, C] bool -> uint8 [R*C] on the device (None if absent)."""
        if self_attn_mask is not None:
            raise NotImplementedError           # same as the reference (modules.py:776-777, 909-910)
        if self_attn_padding_mask is None:
            return None
        m = self_attn_padding_mask
        if m.dim() != 3 or m.shape[0] != 1 or tuple(m.shape[1:]) != (R, C):
            raise ValueError(f"expected padding mask of shape [1, {R}, {C}], got {tuple(m.shape)}")
        return m[0].to(torch.uint8).contiguous().view(-1)


class RowSelfAttention(_AxialAttentionBase):
    """Tied row attention; mirrors modules.py:688-821.  forward(x[R,C,1,D]) -> (out[R,C,1,D], probs[H,1,C,C])."""

    attn_shape = "hnij"

    def align_scaling(self, q: torch.Tensor) -> float:
        return ops.row_scaling(q.size(0))

    def forward(self, x, self_attn_mask=None, self_attn_padding_mask=None, _residual=None):
        _check_inference(self, self.dropout)
        x2, R, C, D = _tokens_2d(x)
        mask = self._mask_bytes(self_attn_mask, self_attn_padding_mask, R, C)
        H = self.num_heads
        res2 = None if _residual is None else _residual.contiguous().view(R * C, D)
        mode = _mode_of(self)
        if mode is not None and not (mask is not None and ops.row_chunks(R, C, self.max_tokens_per_msa)[0]):
            # 16-bit matrix-core mode (the chunked+padded case stays on the exact kernels, like rnamsm_forward)
            split, fmt = mode
            qkv = self._qkv_planes(x2, split, fmt)
            q, k, v = _cols(qkv, 0, D), _cols(qkv, D, 2 * D), _cols(qkv, 2 * D, 3 * D)
            if mask is not None:
                ops.zero_plane_rows(qkv, mask, D)                          # q *= 1 - padding_mask
            partial, _ = ops.row_logits16(q, k, R, C, H, fmt=fmt, scale=self.align_scaling(x))
            probs, p_pl = ops.softmax_rows_planes(partial, split=split, fmt=fmt, plane_scale=4096.0,
                                                  key_mask=None if mask is None else mask[:C])
            ctx = ops.row_apply16(p_pl, v, R, C, H, fmt=fmt, out_scale=1.0 / 4096.0)
            return self._project_out(ctx, res2).view(R, C, 1, D), probs.view(H, 1, C, C)
        # padded tokens: q = 0 (modules.py:767-772); keys whose FIRST-row token is <pad>: logit -10000 (:781-785).
        # Without padding the arithmetic is rnamsm_forward's (one per alignment, whatever computes it): q carries dh^-1/2 and
        # align_scaling's 1/sqrt(R) multiplies the summed logits in K5; with padding the factor stays on q as in the driver.
        canon = mask is None
        qkv = self._qkv(x2, self.scaling if canon else self.align_scaling(x), zero_rows=mask)
        # with padding AND R*C above the token budget the reference sums row chunks that were each filled from their
        # own first row (_batched_forward, modules.py:717-750): reproduced; without padding chunking is a re-ordering
        nchunks, rows_per_chunk = ops.row_chunks(R, C, self.max_tokens_per_msa) if mask is not None else (0, 0)
        if nchunks:
            partial, _ = ops.row_logits(qkv[:, :D], qkv[:, D:2 * D], R, C, H, rows_per_chunk=rows_per_chunk)
            probs = ops.softmax_rows(partial, chunk_pad_mask=mask, rows_per_chunk=rows_per_chunk)
        else:
            partial, _ = ops.row_logits(qkv[:, :D], qkv[:, D:2 * D], R, C, H)
            probs = ops.softmax_rows(partial, key_mask=None if mask is None else mask[:C],
                                     logit_scale=ops.depth_scaling(R) if canon else 1.0)
        ctx = ops.row_apply(probs, qkv[:, 2 * D:], R, C, H)
        out = self._project_out(ctx, res2)
        return out.view(R, C, 1, D), probs.view(H, 1, C, C)


class ColumnSelfAttention(_AxialAttentionBase):
    """Column attention; mirrors modules.py:824-945.  forward(x[R,C,1,D]) -> (out[R,C,1,D], probs[H,C,1,R,R]).

    `out` always comes from the fused kernel, which never forms the probabilities.  The reference returns them
    (modules.py:917-924, 926-945) and so does a stand-alone module (`return_probs=True`, the default): they are then
    materialised by a second, HBM-bound launch (rnamsm_col_attn_probs; H*C*R*R floats -- 1.6 GB per layer at R=256,
    C=512).  MSATransformer builds its layers with `return_probs=False` because the model discards them
    (model.py:390); `probs` is None then."""

    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0.0, max_tokens_per_msa: int = 2 ** 16,
                 return_probs: bool = True):
        super().__init__(embed_dim, num_heads, dropout=dropout, max_tokens_per_msa=max_tokens_per_msa)
        self.return_probs = return_probs

    def forward(self, x, self_attn_mask=None, self_attn_padding_mask=None, _residual=None, _want_probs: bool = True):
        _check_inference(self, self.dropout)
        x2, R, C, D = _tokens_2d(x)
        mask = self._mask_bytes(self_attn_mask, self_attn_padding_mask, R, C)
        H = self.num_heads
        res2 = None if _residual is None else _residual.contiguous().view(R * C, D)
        want = self.return_probs and _want_probs
        mode = _mode_of(self)
        if mode is not None:
            split, fmt = mode
            qkv = self._qkv_planes(x2, split, fmt)
            q, k = _cols(qkv, 0, D), _cols(qkv, D, 2 * D)
            ctx = ops.col_attn16(q, k, _cols(qkv, 2 * D, 3 * D), R, C, H, fmt=fmt,
                                 scale=self.scaling, pad_mask=mask if R > 1 else None)
            probs = ops.col_attn_probs16(q, k, R, C, H, fmt=fmt, scale=self.scaling,
                                         pad_mask=mask if R > 1 else None).view(H, C, 1, R, R) if want else None
            return self._project_out(ctx, res2).view(R, C, 1, D), probs
        if mask is None:
            # no padding: q in log2 units (dh^-1/2 * log2 e from the QKV epilogue) and the kernel whose first pass needs no running
            # maximum -- what rnamsm_forward runs (rnamsm_col_attn_fused_prescaled); the probabilities undo the factor (ln 2)
            qkv = self._qkv(x2, self.scaling * 1.4426950408889634)
            ctx = ops.col_attn(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], R, C, H, prescaled=True)
            probs = ops.col_attn_probs(qkv[:, :D], qkv[:, D:2 * D], R, C, H,
                                       scale=0.6931471805599453).view(H, C, 1, R, R) if want else None
            return self._project_out(ctx, res2).view(R, C, 1, D), probs
        qkv = self._qkv(x2, self.scaling)
        # R == 1 reduces to ctx = v; padded keys get score -10000 (modules.py:911-915)
        ctx = ops.col_attn(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], R, C, H, pad_mask=mask if R > 1 else None)
        probs = ops.col_attn_probs(qkv[:, :D], qkv[:, D:2 * D], R, C, H,
                                   pad_mask=mask if R > 1 else None).view(H, C, 1, R, R) if want else None
        out = self._project_out(ctx, res2)
        return out.view(R, C, 1, D), probs


class FeedForwardNetwork(nn.Module):
    """fc2(GELU_erf(fc1(x))); mirrors modules.py:404-427."""

    def __init__(self, embedding_dim: int, ffn_embedding_dim: int, activation_dropout: float = 0.1,
                 max_tokens_per_msa: int = 2 ** 14):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.ffn_embedding_dim = ffn_embedding_dim
        self.max_tokens_per_msa = max_tokens_per_msa
        self.activation_dropout = activation_dropout
        self.fc1 = nn.Linear(embedding_dim, ffn_embedding_dim)
        self.fc2 = nn.Linear(ffn_embedding_dim, embedding_dim)
        self.gemm_dtype = "f32"
        self._w1_planes, self._w2_planes = _WeightPlanes(), _WeightPlanes()

    def forward(self, x, _residual=None):
        _check_inference(self, self.activation_dropout)
        shape = x.shape
        x2 = x.contiguous().view(-1, shape[-1])
        mode = _mode_of(self)
        if mode is not None:     # hidden activation kept as 16-bit planes between the two GEMMs, as in rnamsm_forward
            split, fmt = mode
            h = ops.linear_planes(_act_planes(x2, split, fmt), self._w1_planes.get(self.fc1.weight, split, fmt),
                                  self.fc1.bias.detach(), act=ACT_GELU_ERF, out_planes=True, fmt=fmt)
            res2 = None if _residual is None else _residual.contiguous().view(-1, shape[-1])
            return ops.linear_planes(h, self._w2_planes.get(self.fc2.weight, split, fmt), self.fc2.bias.detach(),
                                     residual=res2, fmt=fmt).view(shape)
        h = ops.linear(x2, self.fc1.weight.detach(), self.fc1.bias.detach(), act=ACT_GELU_ERF)
        res2 = None if _residual is None else _residual.contiguous().view(-1, shape[-1])
        return ops.linear(h, self.fc2.weight.detach(), self.fc2.bias.detach(), act=ACT_NONE, residual=res2).view(shape)


class NormalizedResidualBlock(nn.Module):
    """x + layer(LayerNorm(x)); mirrors modules.py:369-401.  Around this package's RowSelfAttention / ColumnSelfAttention /
    FeedForwardNetwork the residual add is fused into the layer's last GEMM epilogue; ANY other nn.Module is wrapped as the
    reference wraps it (modules.py:385-401): LayerNorm (rnamsm_layernorm), the layer called on the normalised tensor with the
    caller's arguments, a tuple result split into (x, *rest), the residual added by rnamsm_add."""

    def __init__(self, layer: nn.Module, embedding_dim: int, dropout: float = 0.1):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.layer = layer
        self.dropout = dropout
        self.layer_norm = nn.LayerNorm(self.embedding_dim)

    def forward(self, x, *args, **kwargs):
        _check_inference(self, self.dropout)
        xn = ops.layernorm(x, self.layer_norm.weight.detach(), self.layer_norm.bias.detach(), self.layer_norm.eps)
        if isinstance(self.layer, (RowSelfAttention, ColumnSelfAttention, FeedForwardNetwork)):
            return self.layer(xn, *args, _residual=x, **kwargs)
        outputs = self.layer(xn.view(x.shape), *args, **kwargs)                    # a foreign layer: modules.py:388-401 as written
        if isinstance(outputs, tuple):
            y, *out = outputs
        else:
            y, out = outputs, None
        y = ops.add(x, y.to(torch.float32))
        return (y,) + tuple(out) if out is not None else y


class AxialTransformerLayer(nn.Module):
    """row attention -> column attention -> FFN, each pre-LN + residual; mirrors modules.py:191-267.
    forward(x, need_head_weights=True) -> (x, column_attn [H,C,1,R,R], row_attn [H,1,C,C]) as the reference."""

    def __init__(self, embedding_dim: int = 768, ffn_embedding_dim: int = 3072, num_attention_heads: int = 8,
                 dropout: float = 0.1, attention_dropout: float = 0.1, activation_dropout: float = 0.1,
                 max_tokens_per_msa: int = 2 ** 14, column_attention_probs: bool = True) -> None:
        super().__init__()
        self.embedding_dim = embedding_dim
        self.dropout_prob = dropout
        self.row_self_attention = self.build_residual(
            RowSelfAttention(embedding_dim, num_attention_heads, dropout=dropout, max_tokens_per_msa=max_tokens_per_msa))
        self.column_self_attention = self.build_residual(
            ColumnSelfAttention(embedding_dim, num_attention_heads, dropout=dropout, max_tokens_per_msa=max_tokens_per_msa,
                                return_probs=column_attention_probs))
        self.feed_forward_layer = self.build_residual(
            FeedForwardNetwork(embedding_dim, ffn_embedding_dim, activation_dropout=activation_dropout,
                               max_tokens_per_msa=max_tokens_per_msa))

    def build_residual(self, layer: nn.Module):
        return NormalizedResidualBlock(layer, self.embedding_dim, self.dropout_prob)

    def forward(self, x, self_attn_mask=None, self_attn_padding_mask=None, need_head_weights: bool = False):
        x, row_attn = self.row_self_attention(x, self_attn_mask=self_attn_mask,
                                              self_attn_padding_mask=self_attn_padding_mask)
        # the column probabilities are the second of the three return values (modules.py:253-267): formed only when they
        # are returned, and not at all when the layer was built with column_attention_probs=False (MSATransformer)
        x, column_attn = self.column_self_attention(x, self_attn_mask=self_attn_mask,
                                                    self_attn_padding_mask=self_attn_padding_mask,
                                                    _want_probs=need_head_weights)
        x = self.feed_forward_layer(x)
        if need_head_weights:
            return x, column_attn, row_attn
        return x


class MultiheadAttention(_AxialAttentionBase):
    """Generic 1-D self-attention entry point; mirrors the self-attention path of the reference's
    fairseq-style MultiheadAttention (msm/multihead_attention.py:66-397: q*scaling -> bmm -> key_padding_mask -> softmax
    -> bmm -> out_proj) with its signature and DEFAULTS, and loads its q_proj / k_proj / v_proj / out_proj state_dict.

    forward(query[T,B,E], key, value, key_padding_mask=None, need_weights=True, need_head_weights=False) with key / value
    the same tensor as query -> (attn[T,B,E], weights):
      * need_weights (the reference's default, and what its only caller passes, msm/modules.py:123-131): weights are the
        head-averaged probabilities [B,T,T] (:394-397), or the per-head ones [H,B,T,T] with need_head_weights (:389-393).
        The probabilities are materialised per batch element by the tied-row kernels with one alignment row
        (R := 1, C := T: K4 logits, K5 softmax with the key mask, K6 apply), T <= 1024;
      * need_weights=False: the fused column-attention kernel with R := T and C := B (every batch element attends along
        T; the probabilities never exist), weights = None;
      * key_padding_mask [B,T] (:360-369): masked keys get probability 0.  The reference fills -inf, the kernels -10000
        (exp underflows to 0 in fp32 either way); a batch element whose keys are ALL masked is NaN in the reference
        (softmax over -inf only) -- outputs and weights of such an element are set to NaN here too (_reference_nan);
      * attn_mask [T,T] float (round 5; :353-357): ADDED to the scaled scores of every batch element and head before the key
        padding fill and the softmax (a causal mask is -inf above the diagonal).  Takes the weights route whatever
        need_weights says (the fused kernel never sees the scores): the mask joins the fp32 logits of K4 by one rnamsm_add per
        head; a query row left with no admissible key at all is NaN, as in the reference.
    Round 6 -- the options of the reference's module outside plain self-attention (VERDICT r05 "missing" 3), on the GENERAL route
    (_forward_general): cross-attention (key / value of another length; kdim / vdim), add_bias_kv, add_zero_attn, bias=False,
    incremental_state (prev_key / prev_value / prev_key_padding_mask in the caller's dict under this module's own key, as
    fairseq's FairseqIncrementalState keeps them, :24-58, 273-311, 399-434), static_kv, before_softmax (honoured where the
    reference honours it: on its own route, i.e. with need_head_weights or an incremental state -- its torch.nn.functional route,
    :170-199, ignores the flag).  That route always materialises the probabilities with the tied-row kernels on a square frame of
    max(T, S) positions per batch element (K4 logits, the masks added as ONE fp32 [frame, frame] array with -inf at padded /
    forbidden / out-of-range keys, K5, K6) and runs in exact fp32 whatever gemm_dtype says; T, S <= 1024.
    Raises for dropout in training only."""

    def __init__(self, embed_dim, num_heads, kdim=None, vdim=None, dropout=0.0, bias=True, add_bias_kv=False,
                 add_zero_attn=False, self_attention=False, encoder_decoder_attention=False):
        super().__init__(embed_dim, num_heads, dropout=dropout)
        self.embed_dim = embed_dim
        self.kdim = embed_dim if kdim is None else kdim
        self.vdim = embed_dim if vdim is None else vdim
        self.qkv_same_dim = self.kdim == embed_dim and self.vdim == embed_dim
        # (the flags select where k / v come from on the reference's own route, :228-246; round 5 and before forced self-attention)
        self.self_attention = self_attention
        self.encoder_decoder_attention = encoder_decoder_attention
        if self_attention and not self.qkv_same_dim:
            raise AssertionError("Self-attention requires query, key and value to be of the same size")      # :96-98
        if self.kdim % 32 or self.vdim % 32:
            raise ValueError("kdim / vdim must be multiples of 32 (the GEMM's K tile)")
        if not (self.qkv_same_dim and bias):
            self.k_proj = nn.Linear(self.kdim, embed_dim, bias=bias)
            self.v_proj = nn.Linear(self.vdim, embed_dim, bias=bias)
            self.q_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
            self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self._fusable = self.qkv_same_dim and bias                  # the packed [3E, E] QKV weight exists
        if add_bias_kv:
            self.bias_k = nn.Parameter(torch.zeros(1, 1, embed_dim))
            self.bias_v = nn.Parameter(torch.zeros(1, 1, embed_dim))
        else:
            self.bias_k = self.bias_v = None
        self.add_zero_attn = add_zero_attn
        import uuid
        self._incremental_state_id = str(uuid.uuid4())              # :29-30

    # ---- the caller's incremental_state dict, fairseq layout (:24-58, 447-463)
    def _state_key(self) -> str:
        return f"{self._incremental_state_id}.attn_state"

    def _get_input_buffer(self, incremental_state):
        if incremental_state is None or self._state_key() not in incremental_state:
            return {}
        return incremental_state[self._state_key()]

    def _set_input_buffer(self, incremental_state, buffer):
        if incremental_state is not None:
            incremental_state[self._state_key()] = buffer
        return incremental_state

    def reorder_incremental_state(self, incremental_state, new_order):
        """Beam reordering of the buffered keys / values (:436-452)."""
        buf = self._get_input_buffer(incremental_state)
        for name, val in list(buf.items()):
            if val is None:
                continue
            if self.encoder_decoder_attention and val.size(0) == new_order.size(0):
                break
            buf[name] = val.index_select(0, new_order.to(val.device))
        return self._set_input_buffer(incremental_state, buf)

    @staticmethod
    def _reference_nan(out, weights, kpm, need_head_weights, attn_mask=None):
        """msm/multihead_attention.py:353-371: attn_mask added, masked_fill(-inf), then softmax -- a query left with NO admissible
        key (every key padded, or padded / -inf in its attn_mask row) comes out NaN: attention output row and weights row.  The
        kernels fill padded keys with -10000 and give such a query uniform weights over them; one select per tensor restores
        the reference's values (only when a key mask was passed; no host sync).  out [T, B, E]."""
        masked = kpm.bool()[:, None, :]                                             # [B, 1, T(keys)]
        if attn_mask is not None:
            masked = masked | torch.isneginf(attn_mask)[None]                       # [B, T(queries), T(keys)]
        dead = masked.all(dim=2)                                                    # [B, T] or [B, 1]
        nan = torch.full((), float("nan"), device=out.device, dtype=out.dtype)
        out = torch.where(dead.t()[:, :, None], nan, out)                           # [T | 1, B, 1] over [T, B, E]
        if weights is not None:
            weights = torch.where(dead[None, :, :, None] if need_head_weights else dead[:, :, None], nan, weights)
        return out, weights

    def forward(self, query, key=None, value=None, key_padding_mask=None, incremental_state=None, need_weights=True,
                static_kv=False, attn_mask=None, before_softmax=False, need_head_weights=False):
        own_route = incremental_state is not None or static_kv or need_head_weights          # the reference's manual route (:201 ff.)
        plain = (self._fusable and self.bias_k is None and not self.add_zero_attn and incremental_state is None and not static_kv
                 and (key is None or key is query) and (value is None or value is query) and not (before_softmax and own_route))
        if not plain:
            return self._forward_general(query, key, value, key_padding_mask, incremental_state, need_weights, static_kv, attn_mask,
                                         before_softmax and own_route, need_head_weights, own_route)
        out, weights = self._forward(query, key, value, key_padding_mask, None, need_weights, False, attn_mask, False, need_head_weights)
        if key_padding_mask is not None:
            out, weights = self._reference_nan(out, weights, key_padding_mask.to(out.device), need_head_weights and weights is not None,
                                               None if attn_mask is None else attn_mask.to(out.device))
        return out, weights

    def _forward_general(self, query, key, value, key_padding_mask, incremental_state, need_weights, static_kv, attn_mask,
                         before_softmax, need_head_weights, own_route):
        """Every option of msm/multihead_attention.py:154-397 (class docstring, round 6).  Exact fp32."""
        _check_inference(self, self.dropout)
        if query.dim() != 3 or query.shape[2] != self.embed_dim:
            raise ValueError(f"expected query of shape [T, B, {self.embed_dim}], got {tuple(query.shape)}")
        if need_head_weights:
            need_weights = True
        T, B, E = query.shape
        H, dev = self.num_heads, query.device
        saved = None
        if incremental_state is not None:
            saved = self._get_input_buffer(incremental_state)
            if "prev_key" in saved and static_kv:                    # :203-210: the buffered projections stand for key / value
                assert self.encoder_decoder_attention and not self.self_attention
                key = value = None
        # where k / v come from: the flags on the reference's own route (:228-246), the arguments on its functional route (:170-199)
        if own_route and self.self_attention:
            ksrc = vsrc = query
        elif own_route and self.encoder_decoder_attention:
            ksrc = vsrc = key
        else:
            ksrc, vsrc = (query if key is None and not own_route else key), (query if value is None and not own_route else value)
            if own_route and (ksrc is None) != (vsrc is None):
                raise AssertionError("key and value must both be given")

        def lin(x, layer, scale=1.0):
            x2 = x.contiguous().view(-1, x.shape[-1]).float()
            b = None if layer.bias is None else layer.bias.detach()
            return ops.linear(x2, layer.weight.detach(), b, scale=scale, scale_cols=E if scale != 1.0 else 0).view(x.shape[0], B, E)

        q = lin(query, self.q_proj, self.scaling)                    # (q_proj(query)) * dh^-1/2, :247
        k = None if ksrc is None else lin(ksrc, self.k_proj)
        v = None if vsrc is None else lin(vsrc, self.v_proj)
        am = None if attn_mask is None else attn_mask.to(device=dev, dtype=torch.float32)
        if am is not None and (am.dim() != 2 or am.shape[0] != T):
            raise ValueError(f"expected a float attn_mask of shape [{T}, src_len], got {tuple(attn_mask.shape)}")
        kpm = None if key_padding_mask is None else key_padding_mask.to(dev)

        def grow(mask, rows):                                        # one more admissible key: a zero column (:255-264, 323-336)
            return None if mask is None else torch.cat([mask, mask.new_zeros(rows, 1)], dim=1)

        if self.bias_k is not None:                                  # :249-264
            k = torch.cat([k, self.bias_k.detach().float().expand(1, B, E)])
            v = torch.cat([v, self.bias_v.detach().float().expand(1, B, E)])
            am, kpm = grow(am, T), grow(kpm, B)
        if saved is not None:                                        # :273-311
            def unpack(t):                                           # [B, H, S, dh] -> [S, B, E]
                return t.to(dev).permute(2, 0, 1, 3).reshape(t.shape[2], B, E)
            if saved.get("prev_key") is not None:
                pk, pv = unpack(saved["prev_key"]), unpack(saved["prev_value"])
                k = pk if static_kv else torch.cat([pk, k])
                v = pv if static_kv else torch.cat([pv, v])
            S = k.shape[0]
            prev = saved.get("prev_key_padding_mask")
            if prev is not None and static_kv:                       # _append_prev_key_padding_mask, :399-434
                kpm = prev
            elif prev is not None and kpm is not None:
                kpm = torch.cat([prev.float(), kpm.float()], dim=1)
            elif prev is not None:
                kpm = torch.cat([prev.float(), torch.zeros(B, S - prev.shape[1], device=prev.device)], dim=1)
            elif kpm is not None:
                kpm = torch.cat([torch.zeros(B, S - kpm.shape[1], device=kpm.device), kpm.float()], dim=1)
            saved["prev_key"] = k.view(S, B, H, self.head_dim).permute(1, 2, 0, 3).contiguous()
            saved["prev_value"] = v.view(S, B, H, self.head_dim).permute(1, 2, 0, 3).contiguous()
            saved["prev_key_padding_mask"] = kpm
            self._set_input_buffer(incremental_state, saved)
        if k is None:
            raise AssertionError("no keys: key / value are None and the incremental state holds none")
        if self.add_zero_attn:                                       # :318-336
            k = torch.cat([k, k.new_zeros(1, B, E)])
            v = torch.cat([v, v.new_zeros(1, B, E)])
            am, kpm = grow(am, T), grow(kpm, B)
        S = k.shape[0]
        if am is not None and am.shape[1] != S:
            raise ValueError(f"attn_mask has {am.shape[1]} key columns, the keys number {S}")
        if kpm is not None and tuple(kpm.shape) != (B, S):
            raise AssertionError(f"key_padding_mask must be [{B}, {S}], got {tuple(kpm.shape)}")
        C = max(T, S)
        if C > 1024:
            raise NotImplementedError("the general route materialises [H, frame, frame] with kernels built for frames <= 1024")
        kpm_b = None if kpm is None else kpm.to(device=dev).bool()
        neg = float("-inf")
        ctx = torch.empty(T, B, E, device=dev, dtype=torch.float32)
        probs = torch.empty(B, H, T, S, device=dev, dtype=torch.float32)
        scores = torch.empty(B, H, T, S, device=dev, dtype=torch.float32) if before_softmax else None
        dead = torch.zeros(B, T, dtype=torch.bool, device=dev)       # queries without any admissible key: NaN rows in the reference
        for b in range(B):
            frame = torch.zeros(C, 3 * E, device=dev, dtype=torch.float32)      # q | k | v of this element on the square frame
            frame[:T, :E], frame[:S, E:2 * E], frame[:S, 2 * E:] = q[:, b], k[:, b], v[:, b]
            partial, _ = ops.row_logits(frame[:, :E], frame[:, E:2 * E], 1, C, H)
            add = None
            if am is not None or kpm_b is not None or S < C:
                add = torch.zeros(C, C, device=dev, dtype=torch.float32)
                if am is not None:
                    add[:T, :S] = am
                if kpm_b is not None:
                    add[:, :S].masked_fill_(kpm_b[b][None, :], neg)
                add[:, S:] = neg                                     # frame positions past the last key
                dead[b] = torch.isneginf(add[:T, :S]).all(dim=1)
                for h in range(H):
                    ops.add(partial[0, h], add, out=partial[0, h])
            if before_softmax:
                scores[b] = partial[0, :, :T, :S]
                continue
            p = ops.softmax_rows(partial)                            # [H, C, C]
            c_b = ops.row_apply(p, frame[:, 2 * E:], 1, C, H)        # [C, E]
            ctx[:, b] = c_b[:T]
            probs[b] = p[:, :T, :S]
        if before_softmax:                                           # :358-359
            return scores.view(B * H, T, S), v.view(S, B, H, self.head_dim).permute(1, 2, 0, 3).reshape(B * H, S, self.head_dim)
        ob = None if self.out_proj.bias is None else self.out_proj.bias.detach()
        out = ops.linear(ctx.view(T * B, E), self.out_proj.weight.detach(), ob).view(T, B, E)
        nan = torch.full((), float("nan"), device=dev, dtype=torch.float32)
        out = torch.where(dead.t()[:, :, None], nan, out)
        if not need_weights:
            return out, None
        weights = torch.where(dead[:, None, :, None], nan, probs).permute(1, 0, 2, 3).contiguous()      # [H, B, T, S]  (:389-393)
        if not need_head_weights:
            weights = ops.head_mean(weights)                         # [B, T, S]     (:394-397)
        return out, weights

    def _forward(self, query, key=None, value=None, key_padding_mask=None, incremental_state=None, need_weights=True,
                 static_kv=False, attn_mask=None, before_softmax=False, need_head_weights=False):
        if incremental_state is not None or static_kv:
            raise NotImplementedError("incremental decoding is not implemented")
        if before_softmax:
            raise NotImplementedError("before_softmax is not implemented")
        if (key is not None and key is not query) or (value is not None and value is not query):
            raise NotImplementedError("only self-attention (key = value = query) is implemented")
        if need_head_weights:
            need_weights = True                                         # msm/multihead_attention.py:184-185
        _check_inference(self, self.dropout)
        if query.dim() != 3:
            raise ValueError(f"expected query of shape [T, B, E], got {tuple(query.shape)}")
        T, B, E = query.shape
        H = self.num_heads
        kpm = None
        if key_padding_mask is not None:
            if tuple(key_padding_mask.shape) != (B, T):
                raise ValueError(f"expected key_padding_mask of shape [{B}, {T}], got {tuple(key_padding_mask.shape)}")
            kpm = key_padding_mask.to(device=query.device, dtype=torch.uint8).contiguous()
        am = None
        want_weights = need_weights
        if attn_mask is not None:
            if tuple(attn_mask.shape) != (T, T) or not attn_mask.is_floating_point():
                raise ValueError(f"expected a float attn_mask of shape [{T}, {T}], got {attn_mask.dtype} {tuple(attn_mask.shape)}")
            am = attn_mask.to(device=query.device, dtype=torch.float32).contiguous()
            need_weights = True                                         # the scores must exist for the mask to meet them
        x2 = query.contiguous().view(T * B, E)
        mode = _mode_of(self)
        if mode is not None:
            out, weights = self._forward16(x2, T, B, E, kpm, need_weights, need_head_weights, *mode, attn_mask=am)
            return out, (weights if want_weights else None)
        qkv = self._qkv(x2, self.scaling)                               # [T*B, 3E], token (t, b) = row t*B + b
        if not need_weights:
            # fused: R := T, C := B; its pad mask is indexed like the tokens, [T, B]
            mask = None if kpm is None or T == 1 else kpm.t().contiguous().view(-1)
            ctx = ops.col_attn(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], T, B, H, pad_mask=mask)
            return self._project_out(ctx, None).view(T, B, E), None
        if T > 1024:
            raise NotImplementedError("need_weights materialises [H, T, T] per batch element with kernels built for T <= 1024")
        qkv3 = qkv.view(T, B, 3 * E)
        ctx = torch.empty(T, B, E, device=query.device, dtype=torch.float32)
        probs = torch.empty(B, H, T, T, device=query.device, dtype=torch.float32)
        for b in range(B):                                              # one "alignment" of a single row per element
            qb = qkv3[:, b]                                             # [T, 3E] view, row stride B*3E
            partial, _ = ops.row_logits(qb[:, :E], qb[:, E:2 * E], 1, T, H)
            km = self._add_attn_mask(partial, am, None if kpm is None else kpm[b])
            ops.softmax_rows(partial, out=probs[b], key_mask=km)
            ops.row_apply(probs[b], qb[:, 2 * E:], 1, T, H, out=ctx[:, b])
        out = self._project_out(ctx.view(T * B, E), None).view(T, B, E)
        if not want_weights:
            return out, None
        weights = probs.permute(1, 0, 2, 3)                             # [H, B, T, T]  (:389-393)
        if not need_head_weights:
            weights = ops.head_mean(weights)                            # [B, T, T]     (:394-397)
        return out, weights

    @staticmethod
    def _add_attn_mask(partial, am, key_mask=None):
        """attn_weights += attn_mask (msm/multihead_attention.py:353-357) on K4's fp32 logits [nsplit = 1, H, T, T] of one batch
        element, head by head (rnamsm_add, in place).  Returns the key mask K5 still has to apply.
        With BOTH masks the key padding joins the added mask as -inf (the reference's masked_fill value, :360-369) and K5 gets no
        key mask: K5's own fill is -10000, which is only "minus infinity" next to scores of ordinary size -- under a finite
        large-negative attn_mask (-1e9 block masks) a padded key would otherwise outweigh the admissible ones."""
        if am is None:
            return key_mask
        if partial.shape[0] != 1:
            raise NotImplementedError("attn_mask: the single-row logits are expected in one slab")
        if key_mask is not None:
            am = am.masked_fill(key_mask.bool()[None, :], float("-inf"))
        for h in range(partial.shape[1]):
            ops.add(partial[0, h], am, out=partial[0, h])
        return None

    def _forward16(self, x2, T, B, E, kpm, need_weights, need_head_weights, split, fmt, attn_mask=None):
        """The same two routes on the 16-bit kernels (operands as hi(/lo) planes, q unscaled, fp32 softmax)."""
        H = self.num_heads
        qkv = self._qkv_planes(x2, split, fmt)                          # planes [T*B, 3E]
        if not need_weights:
            mask = None if kpm is None or T == 1 else kpm.t().contiguous().view(-1)
            ctx = ops.col_attn16(_cols(qkv, 0, E), _cols(qkv, E, 2 * E), _cols(qkv, 2 * E, 3 * E), T, B, H, fmt=fmt,
                                 scale=self.scaling, pad_mask=mask)
            return self._project_out(ctx, None).view(T, B, E), None
        if T > 1024:
            raise NotImplementedError("need_weights materialises [H, T, T] per batch element with kernels built for T <= 1024")
        qkv3 = tuple(None if p is None else p.view(T, B, 3 * E) for p in qkv)
        ctx = torch.empty(T, B, E, device=x2.device, dtype=torch.float32)
        probs = torch.empty(B, H, T, T, device=x2.device, dtype=torch.float32)
        for b in range(B):
            qb = tuple(None if p is None else p[:, b] for p in qkv3)   # [T, 3E] plane views, row stride B*3E
            partial, _ = ops.row_logits16(_cols(qb, 0, E), _cols(qb, E, 2 * E), 1, T, H, fmt=fmt, scale=self.scaling)
            km = self._add_attn_mask(partial, attn_mask, None if kpm is None else kpm[b])
            pb, p_pl = ops.softmax_rows_planes(partial, split=split, fmt=fmt, plane_scale=4096.0, key_mask=km)
            probs[b] = pb
            ctx[:, b] = ops.row_apply16(p_pl, _cols(qb, 2 * E, 3 * E), 1, T, H, fmt=fmt, out_scale=1.0 / 4096.0)
        out = self._project_out(ctx.view(T * B, E), None).view(T, B, E)
        weights = probs.permute(1, 0, 2, 3)
        if not need_head_weights:
            weights = ops.head_mean(weights)
        return out, weights
