"""MSATransformer shell over the HIP path; mirrors the live CLI model (reference model.py:258-434).

Same constructor keywords, `forward(tokens[B,R,C], repr_layers, need_head_weights, return_contacts)`
result dict, `max_tokens_per_msa_`, and the 275-key state_dict (strict load of a Lightning
`.ckpt['state_dict']`, RNA_MSM_Inference.py:133-135).  `lm_head.*` and `contact_head.*` parameters are
held so that strict loading works.  `return_contacts=True` runs the contact head kernel (§8 f1).  The LM head
(§8 f4) runs by default, as in the reference (model.py:402: `x = self.lm_head(x)` on every forward; read by
utils/likelihood.py:74,78) -- one [T,768]x[768,768] GEMM + LayerNorm + a 128-row GEMM, < 1 % of a forward.  Callers that
never read "logits" (the CLI, bench.py -- the reference's CLI discards them too) pass `need_logits=False`
(or set `model.compute_logits = False`) and get None.
"""
from __future__ import annotations

import ctypes
import warnings
from typing import Dict, Iterable, List, Optional

import torch
import torch.nn as nn

from . import _lib, ops
from .alphabet import RNAAlphabet
from .modules import AxialTransformerLayer



# A process-wide counter of parameter registrations (torch's global hook: fires on every Module.register_parameter / attribute
# assignment of an nn.Parameter).  MSATransformer._packed_weights keeps its list of Parameter objects for as long as the counter stands.
_PARAM_GENERATION = [0]


def _count_parameter_registration(module, name, param):          # noqa: ARG001
    _PARAM_GENERATION[0] += 1
    return None


torch.nn.modules.module.register_module_parameter_registration_hook(_count_parameter_registration)

class _LMHeadParams(nn.Module):
    """Parameters of the reference's RobertaLMHead (modules.py:303-319); run by MSATransformer.lm_logits."""

    def __init__(self, embed_dim: int, output_dim: int, weight: nn.Parameter):
        super().__init__()
        self.dense = nn.Linear(embed_dim, embed_dim)
        self.layer_norm = nn.LayerNorm(embed_dim)
        self.weight = weight                      # tied to embed_tokens.weight (model.py:328-332)
        self.bias = nn.Parameter(torch.zeros(output_dim))


class _ContactHeadParams(nn.Module):
    """Parameter holder for contact_head.regression.* (modules.py:322-366); executed by ops.contact_head."""

    def __init__(self, in_features: int):
        super().__init__()
        self.regression = nn.Linear(in_features, 1, True)


class MSATransformer(nn.Module):
    def __init__(self, vocab: Optional[RNAAlphabet] = None, optimizer_config=None, contact_train_data=None,
                 embed_dim: int = 768, num_attention_heads: int = 12, num_layers: int = 12,
                 embed_positions_msa: bool = True, dropout: float = 0.1, attention_dropout: float = 0.1,
                 activation_dropout: float = 0.1, max_tokens_per_msa: int = 2 ** 14, max_seqlen: int = 1024,
                 embed_positions_msa_dim: Optional[int] = None, return_col_attentions: bool = False):
        """embed_positions_msa_dim / return_col_attentions: the msm/ variant of the shell (msm/model.py:206-423).  Its
        msa_position_embedding is (1, 1024, 1, embed_positions_msa_dim) -- a per-CHANNEL vector per alignment row when the dim is
        embed_dim (msm/model.py:289-292) instead of RNA-MSM's (1, 1024, 1, 1) scalar; None / 1 = the scalar.  A state_dict whose
        tensor has the other shape is accepted as well (load_state_dict reshapes the parameter first).  return_col_attentions:
        forward(..., need_head_weights=True) also returns result["col_attentions"] [B, L, H, C, R, R] (msm/model.py:404-410),
        computed by rnamsm_col_attn_probs layer by layer; refused above COL_ATTENTIONS_MAX_BYTES (the fused column kernel never
        forms these: 16 GB per MSA at M = 256, L = 512)."""
        super().__init__()
        self.vocab = vocab if vocab is not None else RNAAlphabet()
        self.embed_dim = embed_dim
        self.num_attention_heads = num_attention_heads
        self.num_layers = num_layers
        self.embed_positions_msa = embed_positions_msa
        self.dropout = dropout
        self.attention_dropout = attention_dropout
        self.activation_dropout = activation_dropout
        self.max_tokens_per_msa = max_tokens_per_msa
        self.max_seqlen = max_seqlen
        if not embed_positions_msa:
            raise NotImplementedError("embed_positions_msa=False is not part of the CLI path")

        n_vocab, pad = len(self.vocab), self.vocab.pad_idx
        self.embed_tokens = nn.Embedding(n_vocab, embed_dim, padding_idx=pad)
        if embed_positions_msa_dim not in (None, 1, embed_dim):
            raise ValueError(f"embed_positions_msa_dim must be 1 or embed_dim ({embed_dim}), got {embed_positions_msa_dim}")
        self.msa_position_embedding = nn.Parameter(0.01 * torch.randn(1, 1024, 1, embed_positions_msa_dim or 1), requires_grad=False)
        self.return_col_attentions = bool(return_col_attentions)
        self.layers = nn.ModuleList([
            AxialTransformerLayer(embedding_dim=embed_dim, ffn_embedding_dim=4 * embed_dim,
                                  num_attention_heads=num_attention_heads, dropout=dropout,
                                  attention_dropout=attention_dropout, activation_dropout=activation_dropout,
                                  max_tokens_per_msa=max_tokens_per_msa,
                                  column_attention_probs=False)      # discarded by the model (model.py:390)
            for _ in range(num_layers)])
        self.contact_head = _ContactHeadParams(num_layers * num_attention_heads)
        # LearnedPositionalEmbedding table: max_seqlen + pad_idx + 1 rows (modules.py:277-283)
        self.embed_positions = nn.Embedding(max_seqlen + pad + 1, embed_dim, padding_idx=pad)
        self.emb_layer_norm_before = nn.LayerNorm(embed_dim)
        self.emb_layer_norm_after = nn.LayerNorm(embed_dim)
        self.lm_head = _LMHeadParams(embed_dim, n_vocab, self.embed_tokens.weight)
        self.requires_grad_(False)
        self._pack_key = None
        self._pack = None
        self._plist, self._plist_gen = None, -1          # cached list(self.parameters()) and the generation it was built in
        self._workspace = None
        self._lm_pad = None
        self.compute_logits = True        # model.py:402; forward(need_logits=False) skips the LM head
        # Arithmetic of the contractions: "f32" (exact, default), "f16x3" or "bf16" (include/rnamsm.h) -- for the
        # C++ driver and, through the property below, for every mirror module of the layer-wise path
        self.gemm_dtype = "f32"
        # forward(tokens[B,R,C]) with B > 1 (any arithmetic mode): MSAs are run together (rnamsm_forward_batch) in groups of
        # at most batch_token_budget tokens -- pays below ~8 k tokens per MSA, where a lone forward leaves the chip idle
        self.batch_small_msas = True
        self.batch_token_budget = 32768
        self.check_finite = True          # 16-bit modes, checked_forward_one: non-finite outputs (err bit 2) -> that MSA again on f32
        self._planes = None
        self._folded = None
        self._folded16 = None
        # exact path, MSAs without padding: LayerNorm is applied inside the QKV / fc1 GEMMs (include/rnamsm.h, K1 folded);
        # False keeps the separate LayerNorm launches (same results to fp32 rounding)
        self.fold_layernorm = True

    @property
    def gemm_dtype(self) -> str:
        return self._gemm_dtype

    @gemm_dtype.setter
    def gemm_dtype(self, mode: str) -> None:
        if mode not in _lib.DTYPES:
            raise ValueError(f"gemm_dtype must be one of {tuple(_lib.DTYPES)}, got {mode!r}")
        object.__setattr__(self, "_gemm_dtype", mode)
        for m in self.modules():
            if m is not self and hasattr(m, "gemm_dtype"):
                m.gemm_dtype = mode

    # ------------------------------------------------------------------ reference API
    COL_ATTENTIONS_MAX_BYTES = 8 << 30

    @property
    def row_pos_dim(self) -> int:
        """1 = one scalar per alignment row (RNA-MSM, model.py:293-296); embed_dim = a vector per row (msm/model.py:289-292)."""
        return int(self.msa_position_embedding.shape[-1])

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # a checkpoint of the other shell variant: take its msa_position_embedding shape before the strict shape check
        t = state_dict.get(prefix + "msa_position_embedding")
        if t is not None and tuple(t.shape) != tuple(self.msa_position_embedding.shape) and tuple(t.shape) in (
                (1, 1024, 1, 1), (1, 1024, 1, self.embed_dim)):
            cur = self.msa_position_embedding
            self.msa_position_embedding = nn.Parameter(torch.empty(t.shape, dtype=cur.dtype, device=cur.device), requires_grad=False)
            self._pack_key = None
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        # .to() / .float() / .cuda(): parameters may be REPLACED without a registration (torch writes module._parameters directly when
        # it cannot swap .data in place): the cached parameter list of _packed_weights starts over
        object.__setattr__(self, "_plist", None)
        return super()._apply(fn, *args, **kwargs)

    def max_tokens_per_msa_(self, value: int) -> None:
        """model.py:418-428.  The reference uses it to bound memory by chunking; the HIP kernels tile internally, so
        without padding results are identical for every value.  With padding the reference's chunked row attention fills
        its key mask per row chunk (modules.py:717-750), which is reproduced (rnamsm_row_logits_chunked)."""
        self.max_tokens_per_msa = value
        for layer in self.layers:
            layer.row_self_attention.layer.max_tokens_per_msa = value
            layer.column_self_attention.layer.max_tokens_per_msa = value

    def get_sequence_attention(self, tokens):
        return self(tokens.to(device=self.embed_tokens.weight.device), need_head_weights=True)["row_attentions"]

    def predict_contacts(self, tokens):
        return self(tokens, return_contacts=True)["contacts"]

    # ------------------------------------------------------------------ weight packing for rnamsm_forward
    def _packed_weights(self):
        # Is the packed table still the model's weights?  (data_ptr, version) of every parameter -- over a CACHED list of the Parameter
        # objects: walking the module tree for them (self.parameters()) cost 0.6 ms of host time per forward, a fifth of a lone small
        # alignment's forward.  The list is rebuilt whenever ANY module anywhere registered a parameter since (a global counter bumped
        # by torch's parameter-registration hook), which is the only way a Parameter OBJECT of this model can have been replaced.
        if self._plist is None or self._plist_gen != _PARAM_GENERATION[0]:
            object.__setattr__(self, "_plist", list(self.parameters()))
            object.__setattr__(self, "_plist_gen", _PARAM_GENERATION[0])
        params = self._plist
        key = tuple((p.data_ptr(), p._version) for p in params)
        if key == self._pack_key:
            return self._pack
        dev = self.embed_tokens.weight.device
        if dev.type != "cuda":
            raise _lib.RnamsmError("MSATransformer must be moved to the HIP device (model.to('cuda')): no CPU path exists")
        keep: List[torch.Tensor] = []

        def f(t: torch.Tensor) -> torch.Tensor:
            t = t.detach().to(torch.float32).contiguous()
            keep.append(t)
            return t

        table: List[torch.Tensor] = [
            f(self.embed_tokens.weight), f(self.embed_positions.weight), f(self.msa_position_embedding.reshape(-1)),
            f(self.emb_layer_norm_before.weight), f(self.emb_layer_norm_before.bias),
            f(self.emb_layer_norm_after.weight), f(self.emb_layer_norm_after.bias)]
        for layer in self.layers:
            for blk in (layer.row_self_attention, layer.column_self_attention):
                a = blk.layer
                table += [f(blk.layer_norm.weight), f(blk.layer_norm.bias),
                          f(torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0)),
                          f(torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0)),
                          f(a.out_proj.weight), f(a.out_proj.bias)]
            blk = layer.feed_forward_layer
            table += [f(blk.layer_norm.weight), f(blk.layer_norm.bias), f(blk.layer.fc1.weight), f(blk.layer.fc1.bias),
                      f(blk.layer.fc2.weight), f(blk.layer.fc2.bias)]
        assert len(table) == len(_lib.W_GLOBAL) + self.num_layers * len(_lib.W_LAYER)
        ptrs = (ctypes.c_void_p * len(table))(*[t.data_ptr() for t in table])
        dims = _lib.ModelDims(self.num_layers, self.embed_dim, self.num_attention_heads, 4 * self.embed_dim,
                              self.embed_tokens.num_embeddings, self.embed_positions.num_embeddings,
                              self.vocab.pad_idx, float(self.emb_layer_norm_before.eps), self.row_pos_dim)
        self._pack = (dims, ptrs, keep)
        self._pack_key = key
        return self._pack

    def _weight_planes(self, checked: bool = False):
        """bf16 hi/lo planes of the six GEMM weights per layer (rnamsm_split_bf16), built once per weight version.
        checked: the caller has just called _packed_weights() (its walk over 275 parameters costs 0.7 ms of host time per call)."""
        dims, ptrs, keep = self._pack if checked else self._packed_weights()
        fmt = 1 if self.gemm_dtype == "f16x3" else 0
        if self._planes is not None and self._planes[0] is self._pack_key and self._planes[3] == fmt:
            return self._planes[1]
        ng, nl = len(_lib.W_GLOBAL), len(_lib.W_LAYER)
        slots = [_lib.W_LAYER.index(n) for n in ("row_wqkv", "row_wo", "col_wqkv", "col_wo", "fc1_w", "fc2_w")]
        tensors, addrs = [], []
        for layer in range(self.num_layers):
            for s in slots:
                hi, lo = ops.split_bf16(keep[ng + layer * nl + s], fmt=fmt)
                tensors += [hi, lo]
                addrs += [hi.data_ptr(), lo.data_ptr()]
        arr = (ctypes.c_void_p * len(addrs))(*addrs)
        self._planes = (self._pack_key, arr, tensors, fmt)
        return arr

    def _folded_weights(self, checked: bool = False):
        """LayerNorm folded into the Linear it feeds (rnamsm_ln_fold_weights): per layer {Wg, c, d} for the row QKV, the
        column QKV and fc1, built once per weight version.  rnamsm_forward's `ln_folded` table.  checked: see _weight_planes."""
        dims, ptrs, keep = self._pack if checked else self._packed_weights()
        if self._folded is not None and self._folded[0] is self._pack_key:
            return self._folded[1]
        ng, nl = len(_lib.W_GLOBAL), len(_lib.W_LAYER)
        ix = _lib.W_LAYER.index
        triples = [("row_ln_g", "row_ln_b", "row_wqkv", "row_bqkv"), ("col_ln_g", "col_ln_b", "col_wqkv", "col_bqkv"),
                   ("ffn_ln_g", "ffn_ln_b", "fc1_w", "fc1_b")]
        tensors, addrs = [], []
        for layer in range(self.num_layers):
            base = ng + layer * nl
            for g, b, w, bias in triples:
                out = ops.ln_fold_weights(keep[base + ix(w)], keep[base + ix(bias)], keep[base + ix(g)], keep[base + ix(b)])
                tensors += list(out)
                addrs += [t.data_ptr() for t in out]
        arr = (ctypes.c_void_p * len(addrs))(*addrs)
        self._folded = (self._pack_key, arr, tensors)
        return arr

    def _folded_planes(self, checked: bool = False):
        """The folded LayerNorm weights of the 16-bit modes: per layer {Wg_hi, Wg_lo, c, d} for the row QKV, the column QKV and
        fc1 -- Wg = W * gamma split into the mode's planes, c = the row sums of what the planes hold (the GEMM multiplies
        those, so they are what has to cancel), d = bias + W beta.  rnamsm_forward's `ln_folded16` table.  checked: see _weight_planes."""
        dims, ptrs, keep = self._pack if checked else self._packed_weights()
        fmt = 1 if self.gemm_dtype == "f16x3" else 0
        want_lo = self.gemm_dtype != "bf16"
        if self._folded16 is not None and self._folded16[0] is self._pack_key and self._folded16[3] == (fmt, want_lo):
            return self._folded16[1]
        ng, nl = len(_lib.W_GLOBAL), len(_lib.W_LAYER)
        ix = _lib.W_LAYER.index
        triples = [("row_ln_g", "row_ln_b", "row_wqkv", "row_bqkv"), ("col_ln_g", "col_ln_b", "col_wqkv", "col_bqkv"),
                   ("ffn_ln_g", "ffn_ln_b", "fc1_w", "fc1_b")]
        ht = torch.float16 if fmt == 1 else torch.bfloat16
        tensors, addrs = [], []
        for layer in range(self.num_layers):
            base = ng + layer * nl
            for g, b, w, bias in triples:
                wg32, _, d = ops.ln_fold_weights(keep[base + ix(w)], keep[base + ix(bias)], keep[base + ix(g)], keep[base + ix(b)])
                hi, lo = ops.split_bf16(wg32, want_lo=want_lo, fmt=fmt)
                held = hi.view(ht).double() if lo is None else hi.view(ht).double() + lo.view(ht).double()
                c = held.sum(1).float().contiguous()
                tensors += [hi, lo, c, d]
                addrs += [hi.data_ptr(), 0 if lo is None else lo.data_ptr(), c.data_ptr(), d.data_ptr()]
        arr = (ctypes.c_void_p * len(addrs))(*addrs)
        self._folded16 = (self._pack_key, arr, tensors, (fmt, want_lo))
        return arr

    def _get_workspace(self, nbytes: int, device) -> torch.Tensor:
        if self._workspace is None or self._workspace.numel() < nbytes or self._workspace.device != device:
            self._workspace = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._workspace

    # ------------------------------------------------------------------ LM head (§8 f4)
    def lm_logits(self, features: torch.Tensor) -> torch.Tensor:
        """RobertaLMHead.forward (modules.py:312-319): layer_norm(gelu_erf(dense(x))) @ weight.T + bias.
        features [..., D] (output of emb_layer_norm_after) -> logits [..., vocab].  The projection weight
        (tied to embed_tokens, 12 rows) is zero-padded to one 128-row GEMM tile."""
        lm = self.lm_head
        D, V = self.embed_dim, lm.bias.numel()
        x2 = features.contiguous().view(-1, D)
        key = (lm.weight.data_ptr(), lm.weight._version, lm.bias.data_ptr(), lm.bias._version)
        if self._lm_pad is None or self._lm_pad[0] != key:
            wpad = torch.zeros(128, D, device=x2.device, dtype=torch.float32)
            bpad = torch.zeros(128, device=x2.device, dtype=torch.float32)
            wpad[:V] = lm.weight.detach()
            bpad[:V] = lm.bias.detach()
            self._lm_pad = (key, wpad, bpad)
        _, wpad, bpad = self._lm_pad
        h = ops.linear(x2, lm.dense.weight.detach(), lm.dense.bias.detach(), act=_lib.ACT_GELU_ERF)
        h = ops.layernorm(h, lm.layer_norm.weight.detach(), lm.layer_norm.bias.detach(), lm.layer_norm.eps)
        out = ops.linear(h, wpad, bpad)
        return out[:, :V].contiguous().view(*features.shape[:-1], V)

    # ------------------------------------------------------------------ forward
    def forward_one(self, tokens2d: torch.Tensor, has_padding: Optional[bool] = None,
                    need_repr: bool = True, fold_layernorm: Optional[bool] = None) -> Dict[str, torch.Tensor]:
        """One MSA through the C++ driver (rnamsm_forward): tokens int64 [R, C] on the HIP device ->
        {"row_attn" [NL,H,C,C], "repr" [R,C,D], "emb" [C-1,D], "atp" [NL*H,C-1,C-1]}.
        need_repr=False: only what the CLI writes (emb, atp; bit-identical) -- the last layer then skips the rows the
        outputs do not depend on and "repr" holds alignment row 0 only ([1, C, D]).
        fold_layernorm: None = self.fold_layernorm; False = separate LayerNorm launches for this call.
        The result also carries "err" (int32[1] on the device): bit 0 ERR_INDEX a token / position id out of range, bit 1 ERR_FOLD
        the folded LayerNorm's precondition failed, bit 2 ERR_NONFINITE an emb / atp value is inf or NaN.  This entry point does
        NOT look at it (no host sync) and never falls back: in a 16-bit mode an overflow comes back as inf / NaN outputs with
        bit 2 set, whatever `check_finite` says -- callers must test `out["err"] & ERR_NONFINITE` themselves, or use
        checked_forward_one (one sync per MSA; redoes the MSA on the exact path when check_finite is set)."""
        if self.training:
            raise NotImplementedError("rnamsm implements the inference path only: call .eval()")
        if not tokens2d.is_cuda:
            raise _lib.RnamsmError("tokens must be on the HIP device (no CPU path exists)")
        R, C = tokens2d.shape
        if R > 1024:
            raise RuntimeError(
                "Using model with MSA position embedding trained on maximum MSA "
                f"depth of 1024, but received {R} alignments.")                       # model.py:355-359
        dev = tokens2d.device
        if self.embed_tokens.weight.device != dev:
            raise _lib.RnamsmError(f"tokens are on {dev} but the model is on {self.embed_tokens.weight.device}")
        # the library launches on the calling thread's current device / stream: enter the operands' device (a worker
        # thread, or a model on cuda:1, would otherwise launch on device 0 against pointers of another GPU)
        with torch.cuda.device(dev):
            return self._forward_one_on_device(tokens2d, has_padding, need_repr,
                                               self.fold_layernorm if fold_layernorm is None else fold_layernorm)

    def forward_batch(self, tokens3d: torch.Tensor, has_padding: Optional[bool] = None,
                      fold_layernorm: Optional[bool] = None, true_rows: Optional[torch.Tensor] = None,
                      gemm_dtype: Optional[str] = None) -> Dict[str, torch.Tensor]:
        """B same-shape MSAs (ragged ones padded with <pad>; direct-path mask semantics) through rnamsm_forward_batch, in the
        model's arithmetic mode (`gemm_dtype`: None = self.gemm_dtype): tokens [B,R,C] -> row_attn [B,NL,H,C,C], repr [B,R,C,D],
        emb [B,C-1,D], atp [B,NL*H,C-1,C-1], err int32[1] (bits as in forward_one).  has_padding: None = look at the tokens.
        true_rows (int32 [B] on the device): the real depth of every element of a RAGGED batch -- each MSA's tied logits are
        then scaled by its own depth and the element comes out as its unpadded forward would (forward_ragged); None = the
        reference's batch semantics (padded depth).  Like forward_one this entry point never inspects "err" and never falls back:
        test `err & ERR_NONFINITE` in a 16-bit mode, or call checked_forward_batch."""
        if self.training:
            raise NotImplementedError("inference only (model.eval())")
        if not tokens3d.is_cuda:
            raise _lib.RnamsmError("tokens must be on the HIP device (no CPU path exists)")
        assert tokens3d.ndim == 3
        fold = self.fold_layernorm if fold_layernorm is None else fold_layernorm
        mode = gemm_dtype or self.gemm_dtype
        with torch.cuda.device(tokens3d.device):
            B, R, C = tokens3d.shape
            lib = _lib.load()
            dims, ptrs, _ = self._packed_weights()
            dev = tokens3d.device
            NL, H, D = self.num_layers, self.num_attention_heads, self.embed_dim
            toks = tokens3d.to(torch.int64).contiguous()
            if has_padding is None:
                has_padding = bool((toks == self.vocab.pad_idx).any())
            ws_bytes = lib.rnamsm_forward_batch_workspace_bytes(ctypes.byref(dims), B, R, C)
            ws = self._get_workspace(ws_bytes, dev)
            row_attn = torch.empty(B, NL, H, C, C, device=dev, dtype=torch.float32)
            rep = torch.empty(B, R, C, D, device=dev, dtype=torch.float32)
            emb = torch.empty(B, C - 1, D, device=dev, dtype=torch.float32)
            atp = torch.empty(B, NL * H, C - 1, C - 1, device=dev, dtype=torch.float32)
            err = torch.zeros(1, device=dev, dtype=torch.int32)
            dtype = _lib.DTYPES[mode]
            if dtype != _lib.F32 and mode != self.gemm_dtype:
                raise ValueError("forward_batch: a 16-bit gemm_dtype must be the model's own (the weight planes are built per mode)")
            planes = self._weight_planes(checked=True) if dtype != _lib.F32 else None
            folded = self._folded_weights(checked=True) if (fold and not has_padding and dtype == _lib.F32) else None
            _lib.check(lib.rnamsm_forward_batch(ctypes.byref(dims), ptrs, toks.data_ptr(), B, R, C, ws.data_ptr(), ws.numel(),
                                                row_attn.data_ptr(), rep.data_ptr(), emb.data_ptr(), atp.data_ptr(),
                                                err.data_ptr(), int(has_padding),
                                                None if true_rows is None else true_rows.to(dev, torch.int32).contiguous().data_ptr(),
                                                folded, dtype, planes, torch.cuda.current_stream().cuda_stream))
        return {"row_attn": row_attn, "repr": rep, "emb": emb, "atp": atp, "err": err}

    def checked_forward_batch(self, tokens3d: torch.Tensor, has_padding: Optional[bool] = None,
                              true_rows: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """forward_batch + the error word (see checked_forward_one): index errors raise, a 16-bit batch with a non-finite output
        is redone on the exact path, a failed folded-LayerNorm precondition without the fold."""
        out = self.forward_batch(tokens3d, has_padding, true_rows=true_rows)
        err = int(out["err"].item())
        if err & self.ERR_INDEX:
            raise IndexError("batch: token or position index out of range")
        import warnings
        mode = None
        if (err & self.ERR_NONFINITE) and self.gemm_dtype != "f32" and self.check_finite:
            warnings.warn(f"batch: gemm_dtype={self.gemm_dtype!r} produced non-finite outputs (operand outside the 16-bit range); "
                          "the batch is recomputed on the exact fp32 path")
            mode = "f32"
            out = self.forward_batch(tokens3d, has_padding, true_rows=true_rows, gemm_dtype=mode)
            err = int(out["err"].item())
        if err & self.ERR_FOLD:
            warnings.warn("batch: a token row's mean exceeds 32x its spread; LayerNorm is applied in its own launches for this batch")
            out = self.forward_batch(tokens3d, has_padding, fold_layernorm=False, true_rows=true_rows, gemm_dtype=mode)
        return out

    def forward_packed(self, msas: List[torch.Tensor], fold_layernorm: Optional[bool] = None,
                       need_repr: bool = False, gemm_dtype: Optional[str] = None) -> List[Dict[str, torch.Tensor]]:
        """Alignments of DIFFERENT shapes ([R_b, C_b] int64 tokens, column 0 = <cls>, no <pad>) as ONE token-packed batch
        (rnamsm_forward_packed): the alignments lie back to back on the token axis, nothing is padded.
        Returns per MSA what forward_one(need_repr=False) returns -- emb [C_b-1, D], atp [NL*H, C_b-1, C_b-1], row_attn
        [NL, H, C_b, C_b] (views of the packed outputs; with need_repr also repr [R_b, C_b, D]) and the batch's "err" word.
        gemm_dtype: None = the model's mode; "f32" = the exact path whatever the model's mode.  Exact: every alignment's outputs
        are its own forward's, BIT FOR BIT (round 5).  "bf16" / "f16x3" (must be the model's mode: the weight planes are per mode):
        the Linear layers on the 16-bit matrix cores, attention on the exact descriptor kernels -- equal to the alignment's own
        16-bit forward to the mode's rounding.  Never inspects "err" (bit 3 = ERR_PAD_IN_PACKED: a <pad> inside the batch, whose
        masks this path does not build -- forward_ragged reruns such a batch framed)."""
        if self.training:
            raise NotImplementedError("inference only (model.eval())")
        if not msas or not all(t.is_cuda and t.ndim == 2 for t in msas):
            raise _lib.RnamsmError("forward_packed: a non-empty list of [R, C] token tensors on the HIP device")
        mode = gemm_dtype or self.gemm_dtype
        if mode != "f32" and mode != self.gemm_dtype:
            raise ValueError("forward_packed: a 16-bit gemm_dtype must be the model's own (the weight planes are built per mode)")
        dev = msas[0].device
        fold = self.fold_layernorm if fold_layernorm is None else fold_layernorm
        if fold and mode == "f32":
            # The folded LayerNorm is an alignment's own decision (>= ln_fold_min_tokens tokens; knob "ln_fold" = 1) and one launch
            # set is folded or not as a whole: a list that mixes the two classes goes as two packed batches, one per class, so that
            # every member still comes out as its own forward, bit for bit
            lib = _lib.load()
            thr = int(lib.rnamsm_get_param(b"ln_fold_min_tokens"))
            big = [i for i, t in enumerate(msas) if int(t.shape[0]) * int(t.shape[1]) >= thr]
            if 0 < len(big) < len(msas) and int(lib.rnamsm_get_param(b"ln_fold")) == 1:
                small = [i for i in range(len(msas)) if i not in set(big)]
                parts = [(idx, self.forward_packed([msas[i] for i in idx], fold_layernorm, need_repr, gemm_dtype)) for idx in (big, small)]
                err = parts[0][1][0]["err"] | parts[1][1][0]["err"]
                res = [None] * len(msas)
                for idx, out in parts:
                    for i, item in zip(idx, out):
                        item["err"] = err
                        res[i] = item
                return res
        with torch.cuda.device(dev):
            lib = _lib.load()
            dims, ptrs, _ = self._packed_weights()
            NL, H, D = self.num_layers, self.num_attention_heads, self.embed_dim
            B = len(msas)
            shapes = (ctypes.c_int * (2 * B))(*[int(v) for t in msas for v in t.shape])
            toks = torch.cat([t.to(torch.int64).reshape(-1) for t in msas])
            T = toks.numel()
            ws_bytes = lib.rnamsm_forward_packed_workspace_bytes(ctypes.byref(dims), B, shapes)
            if ws_bytes == 0:
                bad = [tuple(t.shape) for t in msas if not (1 <= t.shape[0] <= 1024 and t.shape[1] >= 2)]
                if any(s_[0] > 1024 for s_ in bad):
                    raise RuntimeError("Using model with MSA position embedding trained on maximum MSA "
                                       f"depth of 1024, but received {max(s_[0] for s_ in bad)} alignments.")      # model.py:355-359
                raise _lib.RnamsmError(f"forward_packed: shapes outside the limits: {bad or [tuple(t.shape) for t in msas]}")
            ws = self._get_workspace(ws_bytes, dev)
            cs = [int(t.shape[1]) for t in msas]
            n_map = [NL * H * c * c for c in cs]
            n_emb = [(c - 1) * D for c in cs]
            n_atp = [NL * H * (c - 1) * (c - 1) for c in cs]
            row_attn = torch.empty(sum(n_map), device=dev, dtype=torch.float32)
            rep = torch.empty(T, D, device=dev, dtype=torch.float32)
            emb = torch.empty(sum(n_emb), device=dev, dtype=torch.float32)
            atp = torch.empty(sum(n_atp), device=dev, dtype=torch.float32)
            err = torch.zeros(1, device=dev, dtype=torch.int32)
            dtype = _lib.DTYPES[mode]
            folded = self._folded_weights(checked=True) if (fold and dtype == _lib.F32) else None
            planes = self._weight_planes(checked=True) if dtype != _lib.F32 else None
            _lib.check(lib.rnamsm_forward_packed(ctypes.byref(dims), ptrs, toks.data_ptr(), B, shapes, ws.data_ptr(), ws.numel(),
                                                 row_attn.data_ptr(), rep.data_ptr(), emb.data_ptr(), atp.data_ptr(), err.data_ptr(),
                                                 folded, dtype, planes, torch.cuda.current_stream().cuda_stream))
        res, om, oe, oa, ot = [], 0, 0, 0, 0
        for b, t in enumerate(msas):
            r, c = int(t.shape[0]), cs[b]
            item = {"emb": emb[oe:oe + n_emb[b]].view(c - 1, D), "atp": atp[oa:oa + n_atp[b]].view(NL * H, c - 1, c - 1),
                    "row_attn": row_attn[om:om + n_map[b]].view(NL, H, c, c), "err": err}
            if need_repr:
                item["repr"] = rep[ot:ot + r * c].view(r, c, D)
            res.append(item)
            om, oe, oa, ot = om + n_map[b], oe + n_emb[b], oa + n_atp[b], ot + r * c
        return res

    ERR_PAD_IN_PACKED = 8

    def forward_ragged_begin(self, msas: List[torch.Tensor], gemm_dtype: Optional[str] = None):
        """First half of forward_ragged(packed=True): ENQUEUE the token-packed batch and return (res, mode) without reading its
        error word -- no host sync, so the caller can enqueue the next group behind it and overlap this group's device-to-host
        copies and file writes with that group's compute (the CLI's pooled path, round 6).  forward_ragged_finish completes it."""
        mode = gemm_dtype or self.gemm_dtype
        return self.forward_packed(msas, gemm_dtype=mode), mode

    def _read_err_behind(self, err: torch.Tensor, after: Optional[torch.cuda.Event]) -> int:
        """The error word of a launch set.  after = an event recorded right behind that launch set: the word is copied on a side
        stream that waits for the EVENT, so the host is not held until what was enqueued afterwards has run too."""
        if after is None:
            return int(err.item())
        dev = err.device
        side = self._err_streams.get(dev) if hasattr(self, "_err_streams") else None
        if side is None:
            if not hasattr(self, "_err_streams"):
                object.__setattr__(self, "_err_streams", {})
            side = self._err_streams[dev] = torch.cuda.Stream(dev)
        host = torch.empty(err.shape, dtype=err.dtype, pin_memory=True)
        side.wait_event(after)
        with torch.cuda.stream(side):
            host.copy_(err, non_blocking=True)
            err.record_stream(side)
        side.synchronize()
        return int(host.reshape(-1)[0])

    def forward_ragged_finish(self, msas: List[torch.Tensor], res, mode: str, after: Optional[torch.cuda.Event] = None):
        """Second half: read the batch's error word (ONE sync; behind `after`, an event recorded right after forward_ragged_begin, when
        given) and act on it as forward_ragged documents -- index errors raise, a 16-bit batch with a non-finite output is redone on
        the exact path, a failed folded-LayerNorm precondition reruns without the fold.  Returns the per-MSA results (`res` itself when
        nothing had to be redone), or None when the batch holds <pad> (forward_ragged then reruns it framed)."""
        import warnings
        err = self._read_err_behind(res[0]["err"], after)
        if err & self.ERR_INDEX:
            raise IndexError("batch: token or position index out of range")
        if (err & self.ERR_NONFINITE) and mode != "f32" and self.check_finite and not (err & self.ERR_PAD_IN_PACKED):
            warnings.warn(f"batch: gemm_dtype={mode!r} produced non-finite outputs (operand outside the 16-bit range); "
                          "the batch is recomputed on the exact fp32 path")
            mode = "f32"
            res = self.forward_packed(msas, gemm_dtype=mode)
            err = int(res[0]["err"].item())
        if (err & self.ERR_FOLD) and not (err & self.ERR_PAD_IN_PACKED):
            warnings.warn("batch: a token row's mean exceeds 32x its spread; LayerNorm is applied in its own launches for this batch")
            res = self.forward_packed(msas, fold_layernorm=False, gemm_dtype=mode)
            err = int(res[0]["err"].item())
        return None if (err & self.ERR_PAD_IN_PACKED) else res

    def forward_ragged(self, msas: List[torch.Tensor], packed: Optional[bool] = None,
                       gemm_dtype: Optional[str] = None) -> List[Dict[str, torch.Tensor]]:
        """Alignments of DIFFERENT shapes ([R_b, C_b] int64 tokens, column 0 = <cls>) in one launch set.  packed (None = True):
        the token-packed batch of forward_packed -- no padding -- in `gemm_dtype` (None = the model's mode; "f32" = exact whatever
        the mode); its error word is read (one sync): index errors raise, a 16-bit batch with a non-finite output is redone on
        the exact path (check_finite), a failed folded-LayerNorm precondition reruns without the fold, a <pad> inside the batch
        reruns it framed.
        Otherwise (packed=False, <pad> present): padded into one [max R, max C] frame and run as a ragged batch
        (rnamsm_forward_batch with true_rows, in the model's mode; padding is computed too).  Returns per MSA what
        forward_one(need_repr=False) returns -- emb [C_b-1, D], atp [NL*H, C_b-1, C_b-1], row_attn [NL, H, C_b, C_b]; packed
        and exact: the MSA's own forward bit for bit; framed or 16-bit: equal to it to fp32 rounding / to the mode's rounding."""
        if packed is None:
            packed = True
        if packed:
            res = self.forward_ragged_finish(msas, *self.forward_ragged_begin(msas, gemm_dtype))
            if res is not None:
                return res
        B = len(msas)
        R = max(int(t.shape[0]) for t in msas)
        C = max(int(t.shape[1]) for t in msas)
        dev = msas[0].device
        frame = torch.full((B, R, C), self.vocab.pad_idx, dtype=torch.int64, device=dev)
        for b, t in enumerate(msas):
            frame[b, :t.shape[0], :t.shape[1]] = t
        depths = torch.tensor([int(t.shape[0]) for t in msas], dtype=torch.int32, device=dev)
        # <pad> from the framing, or already inside an alignment (its masks then follow the reference's padding semantics)
        has_padding = any(tuple(t.shape) != (R, C) for t in msas) or bool((frame == self.vocab.pad_idx).any())
        out = self.checked_forward_batch(frame, has_padding=has_padding, true_rows=depths)
        res = []
        for b, t in enumerate(msas):
            cb = int(t.shape[1])
            res.append({"emb": out["emb"][b, :cb - 1], "atp": out["atp"][b, :, :cb - 1, :cb - 1],
                        "row_attn": out["row_attn"][b, :, :, :cb, :cb], "err": out["err"]})
        return res

    # bits of forward_one's "err" (device int32, OR-ed by the kernels): token / position index out of range; a row whose |mean| is
    # so far above its spread that the folded LayerNorm loses > 5 bits; an emb / atp value that is not finite (K10 looks at
    # every value it packs: in the 16-bit modes an operand outside fp16 range surfaces there as inf / NaN)
    ERR_INDEX, ERR_FOLD, ERR_NONFINITE = 1, 2, 4
    _warned_chunked16 = False

    def checked_forward_one(self, tokens2d: torch.Tensor, has_padding: Optional[bool] = None, need_repr: bool = True,
                            what: str = "MSA") -> Dict[str, torch.Tensor]:
        """forward_one + the error word read back (ONE device sync, the only one): raises IndexError for out-of-range tokens
        like the reference's embedding lookup would; an MSA whose 16-bit-mode outputs are not finite is recomputed on the exact
        path; one that trips the folded LayerNorm's precondition (rnamsm.h, K1 folded) is computed again with separate
        LayerNorm launches -- the caller never sees the difference."""
        return self.finish_forward_one(tokens2d, self.forward_one(tokens2d, has_padding, need_repr), has_padding, need_repr, what)

    def finish_forward_one(self, tokens2d: torch.Tensor, out: Dict[str, torch.Tensor], has_padding: Optional[bool] = None,
                           need_repr: bool = True, what: str = "MSA",
                           after: Optional[torch.cuda.Event] = None) -> Dict[str, torch.Tensor]:
        """Second half of checked_forward_one, for callers that pipeline (the CLI's one-by-one loop, round 6): `out` = what
        forward_one(tokens2d, ...) returned (launches enqueued, nothing read back), `after` = an event recorded right behind those
        launches -- the error word is then read on a side stream behind THAT event, so the forward enqueued since is not waited
        for.  Returns `out` itself when nothing had to be redone."""
        err = self._read_err_behind(out["err"], after)
        if err & self.ERR_INDEX:
            raise IndexError(f"{what}: token or position index out of range")
        import warnings
        mode = None                                   # None = self.gemm_dtype
        if (err & self.ERR_NONFINITE) and self.gemm_dtype != "f32" and self.check_finite:
            # f16x3 / bf16 operands live in 16-bit planes: fp16 overflows above 65504.  The synthetic weights stay far inside;
            # a real checkpoint is not known to, so the MSA is redone on the exact path rather than written out as NaN
            warnings.warn(f"{what}: gemm_dtype={self.gemm_dtype!r} produced non-finite outputs (operand outside the 16-bit "
                          "range); this MSA is recomputed on the exact fp32 path")
            with torch.cuda.device(tokens2d.device):
                out = self._forward_one_on_device(tokens2d, has_padding, need_repr, self.fold_layernorm, gemm_dtype="f32")
            mode = "f32"
            err = int(out["err"].item())                  # the retry's word: every bit is recomputed by it
            if err & self.ERR_INDEX:
                raise IndexError(f"{what}: token or position index out of range")
        if err & self.ERR_FOLD:
            warnings.warn(f"{what}: a token row's mean exceeds 32x its spread; LayerNorm is applied in its own launches for "
                          "this MSA instead of inside the GEMMs")
            with torch.cuda.device(tokens2d.device):
                out = self._forward_one_on_device(tokens2d, has_padding, need_repr, False, gemm_dtype=mode)
            if int(out["err"].item()) & self.ERR_INDEX:
                raise IndexError(f"{what}: token or position index out of range")
        return out

    def _forward_one_on_device(self, tokens2d: torch.Tensor, has_padding: Optional[bool],
                               need_repr: bool = True, fold: bool = True,
                               gemm_dtype: Optional[str] = None) -> Dict[str, torch.Tensor]:
        gemm_dtype = gemm_dtype or self.gemm_dtype
        R, C = tokens2d.shape
        lib = _lib.load()
        dims, ptrs, _ = self._packed_weights()
        dev = tokens2d.device
        NL, H, D = self.num_layers, self.num_attention_heads, self.embed_dim
        toks = tokens2d.to(torch.int64).contiguous()
        if has_padding is None:       # padding_mask = tokens.eq(pad); None when nothing is padded (model.py:346-348)
            has_padding = bool((toks == self.vocab.pad_idx).any())
        # the token budget matters only with padding (the chunked row path fills its key mask per chunk)
        max_tokens = min(int(self.max_tokens_per_msa), 2 ** 31 - 1) if has_padding else 0
        ws_bytes = lib.rnamsm_forward_workspace_bytes(ctypes.byref(dims), R, C, int(has_padding), max_tokens)
        ws = self._get_workspace(ws_bytes, dev)
        row_attn = torch.empty(NL, H, C, C, device=dev, dtype=torch.float32)
        rep = torch.empty(R, C, D, device=dev, dtype=torch.float32)
        emb = torch.empty(C - 1, D, device=dev, dtype=torch.float32)
        atp = torch.empty(NL * H, C - 1, C - 1, device=dev, dtype=torch.float32)
        err = torch.zeros(1, device=dev, dtype=torch.int32)
        dtype = _lib.DTYPES[gemm_dtype]
        if dtype != _lib.F32 and max_tokens and lib.rnamsm_row_chunks(R, C, max_tokens) > 0 and not MSATransformer._warned_chunked16:
            # rnamsm_forward runs such an MSA on the exact path as a whole (csrc/forward.hip): say so once instead of silently
            MSATransformer._warned_chunked16 = True
            warnings.warn(f"gemm_dtype={gemm_dtype!r}: a padded MSA above max_tokens_per_msa ({R} x {C} > {max_tokens}) follows the "
                          "reference's chunked mask semantics, which exist in the exact-fp32 kernels only -- it runs in fp32")
        planes = self._weight_planes(checked=True) if dtype != _lib.F32 else None
        folded = self._folded_weights(checked=True) if (dtype == _lib.F32 and not has_padding and fold) else None
        # the 16-bit modes fold only on request (knob ln_fold = 3: measured neutral there): no tables otherwise
        folded16 = self._folded_planes(checked=True) if (dtype != _lib.F32 and not has_padding and fold
                                             and ops.get_param("ln_fold") == 3) else None
        _lib.check(lib.rnamsm_forward(ctypes.byref(dims), ptrs, toks.data_ptr(), R, C, ws.data_ptr(), ws.numel(),
                                      row_attn.data_ptr(), rep.data_ptr(), emb.data_ptr(), atp.data_ptr(),
                                      err.data_ptr(), int(has_padding), max_tokens, _lib.OUT_REPR if need_repr else 0, dtype, planes,
                                      folded, folded16, torch.cuda.current_stream().cuda_stream))
        pruned = not need_repr and dtype == _lib.F32 and not has_padding and R > 1       # rnamsm_forward's condition
        return {"row_attn": row_attn, "repr": rep[:1] if pruned else rep, "emb": emb, "atp": atp, "err": err}

    def _forward_layerwise(self, tokens2d: torch.Tensor, repr_layers: Iterable[int], has_padding: bool = False,
                           col_attentions: Optional[List[torch.Tensor]] = None):
        """Module-by-module path (same HIP kernels, launched from Python) used when intermediate
        representations -- or the column attention probabilities (col_attentions: a list that receives [1,NL,H,C,R,R]) -- are
        requested."""
        R, C = tokens2d.shape
        D = self.embed_dim
        pmask = (tokens2d == self.vocab.pad_idx)[None] if has_padding else None      # [1, R, C]
        x = ops.embed_ln(tokens2d.to(torch.int64), self.embed_tokens.weight.detach(), self.embed_positions.weight.detach(),
                         self.msa_position_embedding.detach().reshape(-1).contiguous(),
                         self.emb_layer_norm_before.weight.detach(), self.emb_layer_norm_before.bias.detach(),
                         self.vocab.pad_idx, self.emb_layer_norm_before.eps, row_pos_dim=self.row_pos_dim).view(R, C, 1, D)
        reps = {}
        if 0 in repr_layers:
            reps[0] = x.permute(2, 0, 1, 3)
        rows, cols = [], []
        for i, layer in enumerate(self.layers):
            if col_attentions is not None:
                attn = layer.column_self_attention.layer
                keep, attn.return_probs = attn.return_probs, True
                try:
                    x, col_attn, row_attn = layer(x, self_attn_padding_mask=pmask, need_head_weights=True)
                finally:
                    attn.return_probs = keep
                cols.append(col_attn.permute(2, 0, 1, 3, 4))                        # [H,C,1,R,R] -> [1,H,C,R,R] (msm/model.py:383)
            else:
                x, _, row_attn = layer(x, self_attn_padding_mask=pmask, need_head_weights=True)
            rows.append(row_attn.permute(1, 0, 2, 3))                               # [1,H,C,C]
            if (i + 1) in repr_layers and (i + 1) != self.num_layers:
                reps[i + 1] = x.permute(2, 0, 1, 3)
        xf = ops.layernorm(x, self.emb_layer_norm_after.weight.detach(), self.emb_layer_norm_after.bias.detach(),
                           self.emb_layer_norm_after.eps)
        if self.num_layers in repr_layers:
            reps[self.num_layers] = xf.permute(2, 0, 1, 3)
        if col_attentions is not None:
            col_attentions.append(torch.stack(cols, 1))                             # [1,NL,H,C,R,R]
        return reps, torch.stack(rows, 1)                                           # [1,NL,H,C,C]

    def forward(self, tokens, repr_layers=[], need_head_weights=False, return_contacts=False, need_logits=None):
        if return_contacts:
            need_head_weights = True
        assert tokens.ndim == 3
        B, R, C = tokens.shape
        if not tokens.is_cuda:
            raise _lib.RnamsmError("tokens must be on the HIP device (no CPU path exists)")
        # padding_mask is batch-global in the reference (model.py:346-348); masks of un-padded elements are all-false
        has_padding = bool((tokens == self.vocab.pad_idx).any())
        need_logits = self.compute_logits if need_logits is None else need_logits
        # the reference keeps `set(repr_layers)` as given (model.py:369) and fills an entry only when a layer index
        # matches (:371, :393, :400): indices outside 0..num_layers -- negative ones included -- select nothing
        repr_set = set(i for i in repr_layers if 0 <= i <= self.num_layers)
        want = set(repr_set)
        if need_logits:
            repr_set = repr_set | {self.num_layers}
        reps: Dict[int, List[torch.Tensor]] = {i: [] for i in repr_set}
        atts: List[torch.Tensor] = []
        want_cols = need_head_weights and self.return_col_attentions               # msm/model.py:404-410
        cols: Optional[List[torch.Tensor]] = [] if want_cols else None
        if want_cols:
            need = 4 * B * self.num_layers * self.num_attention_heads * C * R * R
            if need > self.COL_ATTENTIONS_MAX_BYTES:
                raise _lib.RnamsmError(f"col_attentions of this input would take {need / 2 ** 30:.1f} GiB "
                                       f"(> {self.COL_ATTENTIONS_MAX_BYTES / 2 ** 30:.0f} GiB): build the model with "
                                       f"return_col_attentions=False or pass fewer / shallower alignments")
        fast = repr_set <= {self.num_layers} and not want_cols
        done = 0
        chunked = has_padding and _lib.load().rnamsm_row_chunks(R, C, min(int(self.max_tokens_per_msa), 2 ** 31 - 1)) > 0
        if fast and B > 1 and not chunked and self.batch_small_msas and not self.training and (
                self.gemm_dtype == "f32" or ops.get_param("attn16") != 0):
            # MSAs of a few thousand tokens, same shape or padded to it: their token-parallel launches are shared
            # (rnamsm_forward_batch); a padded batch above the reference's token budget keeps its per-chunk mask semantics
            # (rnamsm_forward per MSA)
            per = max(1, self.batch_token_budget // (R * C))
            while per > 1 and B - done > 1:
                n = min(per, B - done)
                out = self.checked_forward_batch(tokens[done:done + n], has_padding)
                if self.num_layers in repr_set:
                    reps[self.num_layers].append(out["repr"])
                atts.append(out["row_attn"])
                done += n
        for b in range(done, B):
            if fast:
                out = self.checked_forward_one(tokens[b], has_padding)
                if self.num_layers in repr_set:
                    reps[self.num_layers].append(out["repr"].unsqueeze(0))
                atts.append(out["row_attn"].unsqueeze(0))
            else:
                r, a = self._forward_layerwise(tokens[b], repr_set, has_padding, col_attentions=cols)
                for i in repr_set:
                    reps[i].append(r[i])
                atts.append(a)
        full = {i: torch.cat(v, 0) for i, v in reps.items()}
        logits = self.lm_logits(full[self.num_layers]) if need_logits else None     # model.py:402
        result = {"logits": logits, "representations": {i: v for i, v in full.items() if i in want}}
        if need_head_weights:
            result["row_attentions"] = torch.cat(atts, 0)                           # [B, NL, H, C, C]
            if want_cols:
                result["col_attentions"] = torch.cat(cols, 0)                       # [B, NL, H, C, R, R]
        if return_contacts:                                                         # model.py:412-414
            reg = self.contact_head.regression
            maps = torch.cat(atts, 0)
            result["contacts"] = torch.stack(
                [ops.contact_head(maps[b], reg.weight.detach(), reg.bias.detach()) for b in range(B)], 0)
        return result
