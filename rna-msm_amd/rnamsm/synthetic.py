"""Deterministic synthetic checkpoints and MSAs.

The pretrained checkpoint is not available offline (reference README.md:92), so
parity fixtures and the benchmark both run on weights from this generator.  It
is counter based (splitmix64 of a per-tensor stream id + element index, then
Box-Muller in float64) so the values do not depend on torch/numpy RNG versions,
and -- unlike the reference's ``init_weights`` (model.py:89-101), which zeroes
every bias and makes LayerNorm the identity -- every bias and
LayerNorm affine parameter is non-trivial.

The key set is exactly the reference's 275-key state_dict
(RNA_MSM_Inference.py:133-135 loads it with strict=True).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)

VOCAB_SIZE = 12          # msm/data.py:166-172 "rna language": 4 prepend + 7 standard + <mask>
PAD_IDX = 1
CLS_IDX = 0
RESIDUE_TOKENS = (4, 5, 6, 7, 8, 10)   # A G C U X -


def _fnv1a64(text: str) -> np.uint64:
    h = 0xCBF29CE484222325
    for b in text.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return np.uint64(h)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniform_bits(stream: np.uint64, n: int, lane: int) -> np.ndarray:
    """n 64-bit words for (stream, lane); element i depends only on (stream, lane, i)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(2) + np.uint64(lane)
        return _splitmix64(_splitmix64(idx ^ stream) + stream)


def normal(key: str, seed: int, shape: Tuple[int, ...]) -> np.ndarray:
    """Standard normal float64 array, a pure function of (key, seed, shape)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        stream = _splitmix64(np.array([_fnv1a64(key) ^ np.uint64(seed & 0xFFFFFFFF)], dtype=np.uint64))[0]
    u1 = ((_uniform_bits(stream, n, 0) >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    u2 = ((_uniform_bits(stream, n, 1) >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
    return z.reshape(shape)


def state_dict_spec(embed_dim: int = 768, num_layers: int = 10, vocab_size: int = VOCAB_SIZE,
                    max_seqlen: int = 1024, num_heads: int = 12) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, kind) in the reference's state_dict order; kind selects the distribution."""
    D, F = embed_dim, 4 * embed_dim
    spec: List[Tuple[str, Tuple[int, ...], str]] = [
        ("msa_position_embedding", (1, 1024, 1, 1), "rowpos"),     # model.py:293-296
        ("embed_tokens.weight", (vocab_size, D), "weight"),
    ]
    for i in range(num_layers):
        for blk in ("row_self_attention", "column_self_attention"):
            p = f"layers.{i}.{blk}"
            for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
                spec.append((f"{p}.layer.{proj}.weight", (D, D), "weight"))
                spec.append((f"{p}.layer.{proj}.bias", (D,), "bias"))
            spec.append((f"{p}.layer_norm.weight", (D,), "gamma"))
            spec.append((f"{p}.layer_norm.bias", (D,), "beta"))
        p = f"layers.{i}.feed_forward_layer"
        spec += [
            (f"{p}.layer.fc1.weight", (F, D), "weight"), (f"{p}.layer.fc1.bias", (F,), "bias"),
            (f"{p}.layer.fc2.weight", (D, F), "weight"), (f"{p}.layer.fc2.bias", (D,), "bias"),
            (f"{p}.layer_norm.weight", (D,), "gamma"), (f"{p}.layer_norm.bias", (D,), "beta"),
        ]
    spec += [
        ("contact_head.regression.weight", (1, num_layers * num_heads), "weight"),
        ("contact_head.regression.bias", (1,), "bias"),
        ("embed_positions.weight", (max_seqlen + PAD_IDX + 1, D), "weight"),   # modules.py:277-283
        ("emb_layer_norm_before.weight", (D,), "gamma"), ("emb_layer_norm_before.bias", (D,), "beta"),
        ("emb_layer_norm_after.weight", (D,), "gamma"), ("emb_layer_norm_after.bias", (D,), "beta"),
        ("lm_head.weight", (vocab_size, D), "tied"),               # tied to embed_tokens.weight (model.py:328-332)
        ("lm_head.bias", (vocab_size,), "bias"),
        ("lm_head.dense.weight", (D, D), "weight"), ("lm_head.dense.bias", (D,), "bias"),
        ("lm_head.layer_norm.weight", (D,), "gamma"), ("lm_head.layer_norm.bias", (D,), "beta"),
    ]
    return spec


def make_state_dict(seed: int = 0, embed_dim: int = 768, num_layers: int = 10, num_heads: int = 12,
                    weight_std: float = 0.04, bias_std: float = 0.05, ln_std: float = 0.1,
                    max_seqlen: int = 1024) -> Dict[str, np.ndarray]:
    """float32 numpy state_dict with the reference's key names (SURVEY §5 checkpoint row)."""
    out: Dict[str, np.ndarray] = {}
    for key, shape, kind in state_dict_spec(embed_dim, num_layers, VOCAB_SIZE, max_seqlen, num_heads):
        if kind == "tied":
            out[key] = out["embed_tokens.weight"]
            continue
        z = normal(key, seed, shape)
        if kind == "weight":
            v = weight_std * z
        elif kind == "bias":
            v = bias_std * z
        elif kind == "gamma":
            v = 1.0 + ln_std * z
        elif kind == "beta":
            v = ln_std * z
        elif kind == "rowpos":
            v = 0.01 * z
        else:  # pragma: no cover
            raise ValueError(kind)
        out[key] = np.ascontiguousarray(v.astype(np.float32))
    return out


def make_tokens(num_seqs: int, seq_len: int, msa_index: int = 0) -> np.ndarray:
    """Synthetic MSA tokens int64 [num_seqs, seq_len]: column 0 is <cls>, the rest uniform over
    {A,G,C,U,X,-}; seeded 1234 + msa_index.  No <pad>, so padding_mask is None
    (model.py:346-348)."""
    with np.errstate(over="ignore"):
        stream = _splitmix64(np.array([np.uint64(1234 + msa_index)], dtype=np.uint64))[0]
    bits = _uniform_bits(stream, num_seqs * seq_len, 0)
    table = np.asarray(RESIDUE_TOKENS, dtype=np.int64)
    toks = table[((bits >> np.uint64(33)) % np.uint64(len(table))).astype(np.int64)].reshape(num_seqs, seq_len)
    toks[:, 0] = CLS_IDX
    return toks
