"""ORACLE -- test infrastructure, not product code.

CPU restatement (PyTorch eager, fp32 or fp64) of the reference's axial-attention forward
path, used ONLY as the checker in tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg.  The product path (rna-msm_amd/) never imports this file and fails
loudly when its HIP library is missing.

Pinned: yes -- against fixtures produced by importing the reference itself
(/root/reference, authoring container only) with fully randomised seeded weights; see
tests/golden/make_golden.py and tests/test_oracle_golden.py.  The reference ships no value
tests of its own for this path.

Every function cites the reference lines it restates (paths relative to /root/reference).
Layout: one MSA (B=1) is held as x[R, C, D] (reference: [R, C, B=1, D]); token t = r*C + c.
Padding masks: the direct path (R*C <= max_tokens) masks keys by the MSA's first row; the chunked
path fills -10000 per row chunk using each chunk's own first row (modules.py:727-737) and sums the filled slabs -- both are
restated (row_attention) and pinned (forward_padded_b2.npz, forward_padded_b2_chunked.npz).  A batch is processed one
MSA at a time (B is a loop).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

PAD_IDX = 1
LN_EPS = 1e-5          # nn.LayerNorm default, modules.py:383


def _p(params: Dict[str, torch.Tensor], prefix: str, name: str) -> torch.Tensor:
    return params[f"{prefix}.{name}" if prefix else name]


def layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """nn.LayerNorm over the last dim, eps 1e-5, biased variance (modules.py:383,387).  In 16-bit dtypes the
    reference's nn.LayerNorm keeps its statistics in fp32 and rounds only the result, so the same ATen call is made
    here (the spelled-out form below would round every intermediate and overstate the reference's bf16 drift)."""
    if x.dtype in (torch.bfloat16, torch.float16):
        return F.layer_norm(x, (x.shape[-1],), gamma, beta, LN_EPS)
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + LN_EPS) * gamma + beta


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """nn.Linear: x @ w.T + b, w is [out, in] (b may be None: bias=False)."""
    y = x @ w.t()
    return y if b is None else y + b


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    """nn.GELU() exact-erf form (modules.py:416); 16-bit dtypes go through the same ATen kernel as the reference
    (fp32 inside, one rounding)."""
    if x.dtype in (torch.bfloat16, torch.float16):
        return F.gelu(x)
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def positions_from_tokens(tokens: torch.Tensor) -> torch.Tensor:
    """LearnedPositionalEmbedding.forward index math (modules.py:286-290):
    pos = cumsum(tok != pad) * (tok != pad) + pad_idx, per row."""
    mask = (tokens != PAD_IDX).to(torch.int64)
    return torch.cumsum(mask, dim=-1) * mask + PAD_IDX


def embed(tokens: torch.Tensor, params: Dict[str, torch.Tensor]) -> torch.Tensor:
    """K0: model.py:349-362.  tokens int64 [R, C] -> x [R, C, D] after emb_layer_norm_before."""
    R, C = tokens.shape
    if R > 1024:
        raise RuntimeError(
            "Using model with MSA position embedding trained on maximum MSA "
            f"depth of 1024, but received {R} alignments.")          # model.py:355-359
    x = params["embed_tokens.weight"][tokens]
    x = x + params["embed_positions.weight"][positions_from_tokens(tokens)]
    # (1,1024,1,1): one scalar per alignment row, [R,1,1] broadcast (model.py:293-296); the msm/ variant of the shell
    # holds (1,1024,1,D), a vector per row, [R,1,D] (msm/model.py:289-292, :346) -- the same indexing serves both
    x = x + params["msa_position_embedding"][0, :R]
    x = layer_norm(x, params["emb_layer_norm_before.weight"], params["emb_layer_norm_before.bias"])
    pad = tokens == PAD_IDX
    if bool(pad.any()):                                             # model.py:366-367
        x = x * (1 - pad.unsqueeze(-1).to(x.dtype))
    return x


def row_attention_logits(x: torch.Tensor, params, prefix: str, num_heads: int,
                         scaling: float, pad: Optional[torch.Tensor] = None) -> torch.Tensor:
    """RowSelfAttention.compute_attention_weights (modules.py:752-786).
    x [r, C, D] (a chunk of rows) -> logits [H, C, C] summed over those rows and head_dim.  pad [r, C] bool:
    q is zeroed at padded tokens (:767-772) and keys whose first-row token is padded are filled with -10000 (:781-785)."""
    r, C, D = x.shape
    dh = D // num_heads
    q = linear(x, _p(params, prefix, "q_proj.weight"), _p(params, prefix, "q_proj.bias")) * scaling
    k = linear(x, _p(params, prefix, "k_proj.weight"), _p(params, prefix, "k_proj.bias"))
    if pad is not None:
        q = q * (1 - pad.unsqueeze(-1).to(q.dtype))
    q = q.view(r, C, num_heads, dh)
    k = k.view(r, C, num_heads, dh)
    w = torch.einsum("rihd,rjhd->hij", q, k)
    if pad is not None:
        w = w.masked_fill(pad[0][None, None, :], -10000)
    return w


def row_attention(x: torch.Tensor, params, prefix: str, num_heads: int,
                  max_tokens: Optional[int] = None, pad: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """RowSelfAttention.forward / _batched_forward (modules.py:802-821, 717-750).
    Returns (out [R,C,D], probs [H,C,C]).  With max_tokens set and R*C > max_tokens the row-chunked
    accumulate-then-softmax order of _batched_forward is reproduced (same math, different
    summation order)."""
    R, C, D = x.shape
    dh = D // num_heads
    scaling = (dh ** -0.5) / math.sqrt(R)                          # align_scaling, modules.py:713-715
    if max_tokens is not None and R * C > max_tokens:
        # _batched_forward (modules.py:717-750): every row chunk gets the padding mask of ITS rows, so a chunk's -10000
        # fill comes from the chunk's own first row (:727-737) and the filled slabs are then summed -- with padding this
        # differs from the direct path (a pad on a chunk-starting row masks that key; an all-padded chunk start shifts
        # every logit by -10000 and costs ~1e-3 of fp32 resolution)
        max_rows = max(1, max_tokens // C)                          # modules.py:724
        logits = 0
        for s in range(0, R, max_rows):
            logits = logits + row_attention_logits(x[s:s + max_rows], params, prefix, num_heads, scaling,
                                                    None if pad is None else pad[s:s + max_rows])
    else:
        logits = row_attention_logits(x, params, prefix, num_heads, scaling, pad)
    probs = torch.softmax(logits, dim=-1)                           # modules.py:818 / 739
    v = linear(x, _p(params, prefix, "v_proj.weight"), _p(params, prefix, "v_proj.bias")).view(R, C, num_heads, dh)
    ctx = torch.einsum("hij,rjhd->rihd", probs, v).reshape(R, C, D)   # modules.py:797-798
    out = linear(ctx, _p(params, prefix, "out_proj.weight"), _p(params, prefix, "out_proj.bias"))
    return out, probs


def col_attention(x: torch.Tensor, params, prefix: str, num_heads: int,
                  return_probs: bool = False, col_chunk: int = 64, pad: Optional[torch.Tensor] = None):
    """ColumnSelfAttention.compute_attention_update (modules.py:875-924), no padding mask.
    Column slabs are independent (modules.py:849-873) so they are processed `col_chunk` at a time to
    keep the [H, c, R, R] probabilities small; probs are returned only on request (the reference
    computes and discards them)."""
    R, C, D = x.shape
    dh = D // num_heads
    wv, bv = _p(params, prefix, "v_proj.weight"), _p(params, prefix, "v_proj.bias")
    wo, bo = _p(params, prefix, "out_proj.weight"), _p(params, prefix, "out_proj.bias")
    if R == 1:                                                      # modules.py:882-894
        out = linear(linear(x, wv, bv), wo, bo)
        probs = torch.ones(num_heads, C, 1, 1, dtype=x.dtype, device=x.device) if return_probs else None
        return (out, probs) if return_probs else out
    scaling = dh ** -0.5                                            # modules.py:839
    q = (linear(x, _p(params, prefix, "q_proj.weight"), _p(params, prefix, "q_proj.bias")) * scaling).view(R, C, num_heads, dh)
    k = linear(x, _p(params, prefix, "k_proj.weight"), _p(params, prefix, "k_proj.bias")).view(R, C, num_heads, dh)
    v = linear(x, wv, bv).view(R, C, num_heads, dh)
    ctx = torch.empty(R, C, num_heads, dh, dtype=x.dtype, device=x.device)
    all_probs: List[torch.Tensor] = []
    for s in range(0, C, col_chunk):
        e = min(C, s + col_chunk)
        w = torch.einsum("ichd,jchd->hcij", q[:, s:e], k[:, s:e])   # modules.py:907
        if pad is not None:                                         # modules.py:911-915: padded keys (rows j) of column c
            w = w.masked_fill(pad[:, s:e].t()[None, :, None, :], -10000)
        p = torch.softmax(w, dim=-1)                                # modules.py:917
        ctx[:, s:e] = torch.einsum("hcij,jchd->ichd", p, v[:, s:e])  # modules.py:919
        if return_probs:
            all_probs.append(p)
    out = linear(ctx.reshape(R, C, D), wo, bo)
    if return_probs:
        return out, torch.cat(all_probs, dim=1)
    return out


def ffn(x: torch.Tensor, params, prefix: str, token_chunk: Optional[int] = None) -> torch.Tensor:
    """FeedForwardNetwork.forward (modules.py:423-427): fc2(GELU_erf(fc1(x))).  Tokens are independent, so with
    `token_chunk` the [.., 4D] hidden activation is formed that many tokens at a time (memory only: an fp64 truth at
    M = L = 1024 would otherwise hold 26 GB of it plus temporaries)."""
    def run(xc):
        h = gelu_erf(linear(xc, _p(params, prefix, "fc1.weight"), _p(params, prefix, "fc1.bias")))
        return linear(h, _p(params, prefix, "fc2.weight"), _p(params, prefix, "fc2.bias"))
    if token_chunk is None or x.numel() // x.shape[-1] <= token_chunk:
        return run(x)
    flat = x.reshape(-1, x.shape[-1])
    return torch.cat([run(flat[s:s + token_chunk]) for s in range(0, flat.shape[0], token_chunk)], 0).view(x.shape)


def axial_layer(x: torch.Tensor, params, layer: int, num_heads: int,
                max_tokens: Optional[int] = None, pad: Optional[torch.Tensor] = None,
                ffn_token_chunk: Optional[int] = None, col_probs_out: Optional[List[torch.Tensor]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """AxialTransformerLayer.forward (modules.py:242-267) with each sub-block wrapped as
    NormalizedResidualBlock (modules.py:385-401): x + f(LN(x)); dropout is the identity in eval.
    col_probs_out: a list that receives the layer's column probabilities [H, C, R, R] (what msm/model.py:383 stacks)."""
    base = f"layers.{layer}"
    pre = f"{base}.row_self_attention"
    y, row_probs = row_attention(layer_norm(x, params[f"{pre}.layer_norm.weight"], params[f"{pre}.layer_norm.bias"]),
                                 params, f"{pre}.layer", num_heads, max_tokens, pad)
    x = x + y
    pre = f"{base}.column_self_attention"
    cy = col_attention(layer_norm(x, params[f"{pre}.layer_norm.weight"], params[f"{pre}.layer_norm.bias"]),
                       params, f"{pre}.layer", num_heads, return_probs=col_probs_out is not None, pad=pad)
    if col_probs_out is not None:
        cy, cp = cy
        col_probs_out.append(cp)
    x = x + cy
    pre = f"{base}.feed_forward_layer"
    x = x + ffn(layer_norm(x, params[f"{pre}.layer_norm.weight"], params[f"{pre}.layer_norm.bias"]),
                params, f"{pre}.layer", ffn_token_chunk)
    return x, row_probs


def forward(tokens: torch.Tensor, params: Dict[str, torch.Tensor], num_layers: int = 10,
            num_heads: int = 12, max_tokens: Optional[int] = None,
            layers_to_run: Optional[int] = None, force_mask: bool = False,
            ffn_token_chunk: Optional[int] = None, return_col_attentions: bool = False) -> Dict[str, torch.Tensor]:
    """MSATransformer.forward (model.py:338-416) for one MSA, need_head_weights=True,
    repr_layers=[num_layers]; lm_head / contact head are not on this path.
    tokens int64 [R, C].  Returns representation [R, C, D] (after emb_layer_norm_after,
    model.py:396-401) and row_attentions [num_layers, H, C, C] (model.py:392,409, B squeezed)."""
    assert tokens.ndim == 2
    pad = tokens == PAD_IDX
    pad = pad if (bool(pad.any()) or force_mask) else None          # model.py:346-348
    x = embed(tokens, params)
    rows: List[torch.Tensor] = []
    cols: Optional[List[torch.Tensor]] = [] if return_col_attentions else None
    n = num_layers if layers_to_run is None else layers_to_run
    for i in range(n):
        x, pr = axial_layer(x, params, i, num_heads, max_tokens, pad, ffn_token_chunk, cols)
        rows.append(pr)
    x = layer_norm(x, params["emb_layer_norm_after.weight"], params["emb_layer_norm_after.bias"])
    out = {"representation": x, "row_attentions": torch.stack(rows, 0)}
    if return_col_attentions:                                       # msm/model.py:404-410: [L, H, C, R, R] (B squeezed)
        out["col_attentions"] = torch.stack(cols, 0)
    return out


def pack_outputs(result: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """extract_feat's output section (RNA_MSM_Inference.py:151-166): strip <cls>, keep MSA row 0.
    -> emb [L, D], atp [num_layers*H, L, L] (channel = layer*H + head)."""
    att = result["row_attentions"][..., 1:, 1:]
    L = att.shape[-1]
    emb = result["representation"][0, 1:, :]
    return emb.contiguous(), att.reshape(-1, L, L).contiguous()


def multihead_self_attention(x: torch.Tensor, params, prefix: str, num_heads: int,
                             key_padding_mask: Optional[torch.Tensor] = None, return_weights: bool = False,
                             attn_mask: Optional[torch.Tensor] = None):
    """Self-attention path of msm/multihead_attention.py:154-397 (no bias_kv, eval): q = q_proj(x)*dh^-0.5
    (:256), bmm (:349), attn_mask [T, T] ADDED to the scores of every batch element and head (:353-357), key_padding_mask
    [B, T] -> -inf on masked keys (:360-369), softmax (:371), bmm (:379), out_proj
    (:387).  x [T, B, E] -> out [T, B, E]; with return_weights also the per-head probabilities [H, B, T, T] (:389-393;
    the reference's default returns their mean over heads, :394-397)."""
    T, B, E = x.shape
    dh = E // num_heads
    q = (linear(x, _p(params, prefix, "q_proj.weight"), _p(params, prefix, "q_proj.bias")) * dh ** -0.5).view(T, B, num_heads, dh)
    k = linear(x, _p(params, prefix, "k_proj.weight"), _p(params, prefix, "k_proj.bias")).view(T, B, num_heads, dh)
    v = linear(x, _p(params, prefix, "v_proj.weight"), _p(params, prefix, "v_proj.bias")).view(T, B, num_heads, dh)
    w = torch.einsum("ibhd,jbhd->bhij", q, k)
    if attn_mask is not None:
        w = w + attn_mask.to(w.dtype)[None, None]
    if key_padding_mask is not None:
        w = w.masked_fill(key_padding_mask.to(torch.bool)[:, None, None, :], float("-inf"))
    p = torch.softmax(w, -1)
    ctx = torch.einsum("bhij,jbhd->ibhd", p, v).reshape(T, B, E)
    out = linear(ctx, _p(params, prefix, "out_proj.weight"), _p(params, prefix, "out_proj.bias"))
    return (out, p.transpose(0, 1)) if return_weights else out


def multihead_attention(query: torch.Tensor, key: Optional[torch.Tensor], value: Optional[torch.Tensor], w: Dict[str, torch.Tensor],
                        num_heads: int, key_padding_mask: Optional[torch.Tensor] = None,
                        attn_mask: Optional[torch.Tensor] = None, add_zero_attn: bool = False,
                        saved: Optional[Dict[str, Optional[torch.Tensor]]] = None, static_kv: bool = False,
                        before_softmax: bool = False):
    """The GENERAL form of msm/multihead_attention.py:154-397 in eval mode -- everything the self-attention restatement above leaves
    out: cross-attention (key / value of another length and width, :228-246), bias_k / bias_v (one learned key / value appended to
    every batch element, :249-264), add_zero_attn (an all-zero key / value appended, :318-336), and incremental decoding through
    a saved state (:273-311: prev_key / prev_value [B, H, S, dh] are prepended -- or, static_kv, replace the new ones -- and the
    padding masks are joined by _append_prev_key_padding_mask, :399-434).
    w: q_proj / k_proj / v_proj / out_proj .weight (+ .bias, optional), bias_k / bias_v [1, 1, E] (optional).
    query [T, B, E]; key [S, B, kdim] / value [S, B, vdim] (None with static_kv once the state holds them).
    Returns (out [T, B, E], per-head probabilities [H, B, T, S_total]); with `saved` given the dict is updated in place as the
    reference updates its buffer.  before_softmax: (masked scores [B*H, T, S_total], v [B*H, S_total, dh]) as :371-372."""
    T, B, E = query.shape
    H, dh = num_heads, E // num_heads

    def proj(x, name):
        return linear(x, w[f"{name}.weight"], w.get(f"{name}.bias"))

    q = proj(query, "q_proj") * dh ** -0.5                                                   # :247
    k = None if key is None else proj(key, "k_proj")
    v = None if value is None else proj(value, "v_proj")
    am, kpm = attn_mask, key_padding_mask
    if w.get("bias_k") is not None:                                                          # :249-264
        k = torch.cat([k, w["bias_k"].expand(1, B, E)])
        v = torch.cat([v, w["bias_v"].expand(1, B, E)])
        if am is not None:
            am = torch.cat([am, am.new_zeros(am.shape[0], 1)], dim=1)
        if kpm is not None:
            kpm = torch.cat([kpm, kpm.new_zeros(B, 1)], dim=1)
    if saved is not None:                                                                    # :273-311
        def from_state(t):                                                                   # [B, H, S, dh] -> [S, B, E]
            return t.permute(2, 0, 1, 3).reshape(t.shape[2], B, E)
        if saved.get("prev_key") is not None:
            pk, pv = from_state(saved["prev_key"]), from_state(saved["prev_value"])
            k = pk if static_kv else torch.cat([pk, k])
            v = pv if static_kv else torch.cat([pv, v])
        prev = saved.get("prev_key_padding_mask")
        S = k.shape[0]
        if prev is not None and static_kv:                                                   # :399-434
            kpm = prev
        elif prev is not None and kpm is not None:
            kpm = torch.cat([prev.float(), kpm.float()], dim=1)
        elif prev is not None:
            kpm = torch.cat([prev.float(), torch.zeros(B, S - prev.shape[1])], dim=1)
        elif kpm is not None:
            kpm = torch.cat([torch.zeros(B, S - kpm.shape[1]), kpm.float()], dim=1)
        saved["prev_key"] = k.view(S, B, H, dh).permute(1, 2, 0, 3).contiguous()
        saved["prev_value"] = v.view(S, B, H, dh).permute(1, 2, 0, 3).contiguous()
        saved["prev_key_padding_mask"] = kpm
    if add_zero_attn:                                                                        # :318-336
        k = torch.cat([k, k.new_zeros(1, B, E)])
        v = torch.cat([v, v.new_zeros(1, B, E)])
        if am is not None:
            am = torch.cat([am, am.new_zeros(am.shape[0], 1)], dim=1)
        if kpm is not None:
            kpm = torch.cat([kpm, kpm.new_zeros(B, 1)], dim=1)
    S = k.shape[0]
    scores = torch.einsum("ibhd,jbhd->bhij", q.view(T, B, H, dh), k.view(S, B, H, dh))        # :338
    if am is not None:
        scores = scores + am.to(scores.dtype)[None, None]                                    # :343-347
    if kpm is not None:
        scores = scores.masked_fill(kpm.to(torch.bool)[:, None, None, :], float("-inf"))     # :349-356
    if before_softmax:
        return scores.reshape(B * H, T, S), v.view(S, B, H, dh).permute(1, 2, 0, 3).reshape(B * H, S, dh)
    p = torch.softmax(scores, -1)
    ctx = torch.einsum("bhij,jbhd->ibhd", p, v.view(S, B, H, dh)).reshape(T, B, E)
    return linear(ctx, w["out_proj.weight"], w.get("out_proj.bias")), p.transpose(0, 1)


def lm_head(features: torch.Tensor, params: Dict[str, torch.Tensor]) -> torch.Tensor:
    """RobertaLMHead.forward (modules.py:312-319): dense -> gelu (erf form, modules.py:11-20) -> layer_norm ->
    tied projection + bias.  features [..., D] -> logits [..., vocab]."""
    x = linear(features, params["lm_head.dense.weight"], params["lm_head.dense.bias"])
    x = gelu_erf(x)
    x = layer_norm(x, params["lm_head.layer_norm.weight"], params["lm_head.layer_norm.bias"])
    return x @ params["lm_head.weight"].t() + params["lm_head.bias"]


def contact_head(row_attentions: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """ContactPredictionHead.forward (modules.py:344-366) for the RNA alphabet (prepend_bos, no eos) with
    symmetrize / apc (utils/tensor.py:98-113).  row_attentions [NL, H, C, C] -> contacts [C-1, C-1]."""
    a = row_attentions[..., 1:, 1:]
    T = a.shape[-1]
    a = a.reshape(-1, T, T)
    a = a + a.transpose(-1, -2)
    a1 = a.sum(-1, keepdim=True)
    a2 = a.sum(-2, keepdim=True)
    a12 = a.sum((-1, -2), keepdim=True)
    a = a - (a1 * a2) / a12
    z = torch.einsum("cij,c->ij", a, weight.reshape(-1)) + bias.reshape(())
    return torch.sigmoid(z)


def to_torch_params(state: Dict[str, "object"], dtype=torch.float32, device="cpu") -> Dict[str, torch.Tensor]:
    """numpy / torch state_dict -> torch tensors of `dtype` on `device` (CPU by default; the tests also run this same
    code in fp64 on the GPU box's device as the full-size truth, tests/truth.py -- still only as the checker)."""
    out = {}
    for k, v in state.items():
        t = v if isinstance(v, torch.Tensor) else torch.from_numpy(v)
        out[k] = t.detach().to(device, dtype)
    return out
