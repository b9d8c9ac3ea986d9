"""Masked pseudo-likelihood scoring on top of the HIP forward (§8 f4: the reference's utils/likelihood.py).

`sequence_logits` mirrors utils/likelihood.py:39-82 for alignment input: position i of the query (row 0) is replaced by
<mask> in copy i of the MSA, every copy runs the 10-layer forward and the LM head, and the logits at the masked position
are collected.  The copies go through `MSATransformer.forward` in batches of max(1, max_tokens // (R*C)) MSAs, as the
reference does; every FLOP is in librnamsm_hip.so.
"""
from __future__ import annotations

from typing import Optional, Sequence, Union

import numpy as np
import torch

IntSeq = Union[Sequence[int], np.ndarray, torch.Tensor]


def mask_and_repeat(tokens: torch.Tensor, indices: IntSeq, mask_idx: int) -> torch.Tensor:
    """tokens [R, C] -> [len(indices), R, C] with copy i masked at (row 0, indices[i])  (utils/likelihood.py:18-27)."""
    idx = torch.as_tensor(indices, dtype=torch.long, device=tokens.device)
    out = tokens.unsqueeze(0).repeat(len(idx), 1, 1)
    out[torch.arange(len(idx), device=tokens.device), 0, idx] = mask_idx
    return out


@torch.no_grad()
def sequence_logits(model, tokens: torch.Tensor, mask_positions: bool = True, max_tokens: int = 2 ** 14,
                    indices: Optional[IntSeq] = None) -> torch.Tensor:
    """tokens int64 [R, C] (or [C]: a one-row alignment) on the HIP device, column 0 = <cls>.
    Returns logits [len(indices), vocab] at the query positions `indices` (default: every column after <cls>)."""
    if tokens.dim() == 1:
        tokens = tokens[None]
    assert tokens.dim() == 2
    vocab = model.vocab
    R, C = tokens.shape
    if indices is None:
        indices = torch.arange(int(vocab.prepend_bos), C - int(vocab.append_eos))
    idx = torch.as_tensor(indices, dtype=torch.long, device=tokens.device)
    n = len(idx)
    if not mask_positions:                                                   # utils/likelihood.py:80-81
        out = model(tokens[None], need_logits=True)["logits"]                # [1, R, C, V]
        return out[0, 0, idx]
    masked = mask_and_repeat(tokens, idx, vocab.mask_idx)
    logits = torch.zeros(n, len(vocab), device=tokens.device)
    batch = max(1, max_tokens // (R * C))
    for s in range(0, n, batch):
        out = model(masked[s:s + batch], need_logits=True)["logits"]         # [b, R, C, V]
        b = out.shape[0]
        logits[s:s + b] = out[torch.arange(b, device=out.device), 0, idx[s:s + b]]
    return logits


@torch.no_grad()
def masked_marginal_scores(model, tokens: torch.Tensor, mask_positions: bool = True, max_tokens: int = 2 ** 14,
                           indices: Optional[IntSeq] = None) -> torch.Tensor:
    """log p(token | masked context) - log p(wild type | masked context) per query position and vocabulary entry
    (utils/likelihood.py:86-120 up to its final re-mapping onto the 20 protein letters, which has no RNA counterpart):
    [len(indices), vocab]; the wild-type column is 0."""
    if tokens.dim() == 1:
        tokens = tokens[None]
    vocab = model.vocab
    C = tokens.shape[1]
    if indices is None:
        indices = torch.arange(int(vocab.prepend_bos), C - int(vocab.append_eos))
    idx = torch.as_tensor(indices, dtype=torch.long, device=tokens.device)
    lp = sequence_logits(model, tokens, mask_positions, max_tokens, idx).log_softmax(-1)
    wt = tokens[0, idx]
    return lp - lp[torch.arange(len(idx), device=lp.device), wt].unsqueeze(1)
