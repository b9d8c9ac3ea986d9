"""a2m / FASTA alignment reader and row sub-sampling for the CLI path.

Replaces MSA.from_fasta (utils/align.py:291-317) + A2MDataset.__getitem__ (dataset.py:80-92) for this
path.  Row sub-sampling: the reference defaults to the external `hhfilter` binary (utils/align.py:68-102),
which is not available offline; supported here are `first` (keep the first N rows), the reference's weighted random
`sample-pretrained` (utils/align.py:150-163; weights on the device, the draw by numpy's generator) and its
greedy `diversity-max` / `diversity-min` (utils/align.py:128-148), on the HIP device when one is given
(rnamsm_greedy_select) or on the host -- index-identical to the reference either way.
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Tuple, Union

import numpy as np

from .alphabet import RNAAlphabet

SAMPLE_METHODS = ("hhfilter", "sample-pretrained", "diversity-max", "diversity-min", "first")


def read_fasta_records(path_or_text: Union[str, Path], is_text: bool = False) -> List[Tuple[str, str]]:
    """(description, sequence) records the way the reference's parser (Bio.SeqIO "fasta" = SimpleFastaParser) yields them:
    lines split at newlines only, a record's lines right-stripped and joined, then blanks and carriage returns removed
    (interior tabs stay and are invalid tokens, as there); text before the first '>' is skipped."""
    text = path_or_text if is_text else Path(path_or_text).read_text()
    records: List[Tuple[str, str]] = []
    header, parts = None, []

    def close():
        records.append((header, "".join(parts).replace(" ", "").replace("\r", "")))

    for raw in text.split("\n"):
        if raw[:1] == ">":
            if header is not None:
                close()
            header, parts = raw[1:].rstrip(), []
        elif header is not None:
            parts.append(raw.rstrip())
    if header is not None:
        close()
    return records


def greedy_select(tokens: np.ndarray, num_seqs: int, mode: str = "max") -> np.ndarray:
    """Row indices (sorted) chosen by the reference's greedy max/min mean-Hamming rule
    (utils/align.py:128-148): start from row 0, repeatedly add the row whose mean normalised Hamming
    distance to the rows chosen so far is largest (smallest), first index on ties.

    Candidates with equal mismatch totals are everywhere in real alignments (at step n the totals are integers below
    n*L), and since m/L is inexact in float64 the winner among them is decided by the ORDER of the n additions.  The
    reference takes `.mean(0)` of a [n, candidates] matrix whose reduction axis is the contiguous one (np.delete along
    axis 1 hands back such a layout), i.e. numpy's pairwise summation: 8 interleaved accumulators up to 128 terms, halves
    rounded to multiples of 8 above.  Keeping one float64 per (candidate, step) and reducing along the contiguous step
    axis reproduces exactly that order, so the indices are the reference's (pinned on the shipped 1176-row 2DRB_1
    alignment, where a running sum goes a different way at step 46)."""
    depth = tokens.shape[0]
    if depth <= num_seqs:
        return np.arange(depth)
    body = tokens[:, 1:] if tokens.shape[1] > 1 else tokens
    pick = np.argmax if mode == "max" else np.argmin
    chosen = [0]
    taken = np.zeros(depth, dtype=bool)
    taken[0] = True
    hist = np.empty((depth, num_seqs - 1), dtype=np.float64)        # [candidate, step]: step axis contiguous
    for step in range(1, num_seqs):
        last = body[chosen[-1]]
        hist[:, step - 1] = (body != last[None, :]).mean(1)         # cdist(..., "hamming") = mismatches / L
        cand = np.flatnonzero(~taken)
        best = cand[pick(hist[cand, :step].sum(1) / step)]
        chosen.append(int(best))
        taken[best] = True
    return np.array(sorted(chosen))


def greedy_select_device(tokens: np.ndarray, num_seqs: int, mode: str, device) -> np.ndarray:
    """greedy_select on the HIP device (rnamsm_greedy_select): same indices as the host version, O(N*L) bytes per
    step at HBM/L2 speed instead of numpy speed -- for alignments with 1e4..1e5 rows."""
    import torch
    from . import ops
    if tokens.shape[0] <= num_seqs:
        return np.arange(tokens.shape[0])
    body = tokens[:, 1:] if tokens.shape[1] > 1 else tokens
    u8 = torch.from_numpy(np.ascontiguousarray(body.astype(np.uint8))).to(device)
    return ops.greedy_select(u8, num_seqs, mode).cpu().numpy().astype(np.int64)


def msa_weights(tokens: np.ndarray, seqid_cutoff: float = 0.2, device=None) -> np.ndarray:
    """MSA.weights (utils/align.py:250-253): 1 / number of rows within `seqid_cutoff` normalised Hamming distance (the
    row itself included), float64 [N].  The distances are over the alignment columns (no <cls>); with `device` the O(N^2 L)
    comparison runs on the GPU (rnamsm_msa_weights) -- same integers, same float64 comparison, identical weights."""
    body = np.ascontiguousarray((tokens[:, 1:] if tokens.shape[1] > 1 else tokens).astype(np.uint8))
    if device is not None:
        import torch
        from . import ops
        return ops.msa_weights(torch.from_numpy(body).to(device), seqid_cutoff).cpu().numpy()
    n = body.shape[0]
    counts = np.zeros(n, dtype=np.int64)
    step = max(1, (1 << 24) // max(1, n * body.shape[1]))
    for i0 in range(0, n, step):
        dist = (body[i0:i0 + step, None, :] != body[None, :, :]).mean(-1)       # pdist(.., "hamming") = mismatches / L
        counts[i0:i0 + step] = (dist < seqid_cutoff).sum(1)
    return 1 / counts


def sample_weights(tokens: np.ndarray, num_seqs: int, rng=None, seqid_cutoff: float = 0.2, device=None) -> np.ndarray:
    """Row indices drawn like MSA.sample_weights (utils/align.py:150-163): row 0 plus num_seqs - 1 of the others without
    replacement with probability proportional to their sequence weight, ascending.  `rng`: a numpy RandomState; None =
    numpy's global one, which is what the reference draws from (seeded by seed_everything(42), RNA_MSM_Inference.py:17)."""
    depth = tokens.shape[0]
    if depth <= num_seqs:
        return np.arange(depth)
    rng = np.random if rng is None else rng
    w = msa_weights(tokens, seqid_cutoff, device)[1:]
    w = w / w.sum()
    idx = rng.choice(depth - 1, size=num_seqs - 1, replace=False, p=w) + 1
    return np.append(0, np.sort(idx))


def load_msa_tokens(path: Union[str, Path], alphabet: RNAAlphabet, max_seqs_per_msa: int = 512,
                    sample_method: str = "hhfilter", device=None, rng=None) -> np.ndarray:
    """.a2m_msa2 file -> int64 tokens [R <= max_seqs, L+1].  With `device` (a HIP device) the sub-sampling arithmetic
    (greedy selection, sequence weights) runs on the GPU.  `rng` feeds `sample-pretrained` (see sample_weights)."""
    if sample_method not in SAMPLE_METHODS:
        raise AssertionError(f"unknown sample_method {sample_method!r}")
    records = read_fasta_records(path)
    tokens = alphabet.encode_a2m_records([seq for _, seq in records])
    if max_seqs_per_msa is None or tokens.shape[0] <= max_seqs_per_msa:
        return tokens
    if sample_method == "first":
        return tokens[:max_seqs_per_msa]
    if sample_method in ("diversity-max", "diversity-min"):
        mode = sample_method.split("-")[1]
        if device is not None:
            return tokens[greedy_select_device(tokens, max_seqs_per_msa, mode, device)]
        return tokens[greedy_select(tokens, max_seqs_per_msa, mode)]
    if sample_method == "sample-pretrained":
        return tokens[sample_weights(tokens, max_seqs_per_msa, rng, device=device)]
    raise NotImplementedError(
        f"sample_method={sample_method!r} needs the external hhfilter binary of the reference (utils/align.py:68-102); "
        "pre-filter the alignment or pass data.sample_method=first | diversity-max | diversity-min | sample-pretrained")
