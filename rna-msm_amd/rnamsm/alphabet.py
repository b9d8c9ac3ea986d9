"""RNA alphabet + tokenizer of the "rna language" architecture.

Replaces msm/data.py:166-172 (Alphabet.from_architecture), msm/constants.py:10-12 and
utils/tokenization.py:107-129,151-165 (Vocab.encode) for this path.  Token ids (bit-exact
contract): <cls>0 <pad>1 <eos>2 <unk>3 A4 G5 C6 U7 X8 N9 -10 <mask>11; sequences get <cls>
prepended and no <eos>.

Implementation differs from the reference's digitize-over-sorted-symbols: one 256-entry byte
lookup table maps an a2m character straight to its token id, with the a2m clean-up rules of
utils/align.py:309-312 folded in (T->U, IUPAC ambiguity codes -> X, lowercase / '.' / '*'
insertions dropped).
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np

_DROP, _INVALID = -2, -1


class RNAAlphabet:
    prepend_toks = ("<cls>", "<pad>", "<eos>", "<unk>")
    standard_toks = ("A", "G", "C", "U", "X", "N", "-")
    append_toks = ("<mask>",)
    prepend_bos = True
    append_eos = False
    use_msa = True

    def __init__(self):
        self.all_toks = list(self.prepend_toks) + list(self.standard_toks) + list(self.append_toks)
        self.tok_to_idx: Dict[str, int] = {t: i for i, t in enumerate(self.all_toks)}
        self.cls_idx = self.bos_idx = self.tok_to_idx["<cls>"]
        self.pad_idx = self.padding_idx = self.tok_to_idx["<pad>"]
        self.eos_idx = self.tok_to_idx["<eos>"]
        self.unk_idx = self.tok_to_idx["<unk>"]
        self.mask_idx = self.tok_to_idx["<mask>"]
        # raw a2m byte -> token id (clean-up rules folded in)
        lut = np.full(256, _INVALID, dtype=np.int64)
        for ch in "abcdefghijklmnopqrstuvwxyz.*":
            lut[ord(ch)] = _DROP
        for ch in "AGCUX-":
            lut[ord(ch)] = self.tok_to_idx[ch]
        lut[ord("T")] = self.tok_to_idx["U"]
        for ch in "RYKMSWBDHVN":
            lut[ord(ch)] = self.tok_to_idx["X"]
        self._a2m_lut = lut
        # already-clean sequence byte -> token id (what Vocab.encode accepts: the single-character tokens)
        clean = np.full(256, _INVALID, dtype=np.int64)
        for ch in self.standard_toks:
            clean[ord(ch)] = self.tok_to_idx[ch]
        self._clean_lut = clean

    @classmethod
    def from_architecture(cls, name: str = "rna language") -> "RNAAlphabet":
        if name != "rna language":
            raise ValueError("Unknown architecture selected")
        return cls()

    def __len__(self) -> int:
        return len(self.all_toks)

    def get_idx(self, tok: str) -> int:
        return self.tok_to_idx.get(tok, self.unk_idx)

    def get_tok(self, ind: int) -> str:
        return self.all_toks[ind]

    def to_dict(self) -> Dict[str, int]:
        return dict(self.tok_to_idx)

    # ------------------------------------------------------------------ encoding
    def _finish(self, rows: Sequence[np.ndarray]) -> np.ndarray:
        if not rows:
            raise ValueError("empty alignment")
        width = len(rows[0])
        assert all(len(r) == width for r in rows), "Seqlen Mismatch!"       # utils/align.py:28-30
        body = np.stack(rows, 0) if width else np.zeros((len(rows), 0), dtype=np.int64)
        if (body == _INVALID).any():
            raise ValueError("Invalid tokens in input")                      # utils/tokenization.py:154-155
        out = np.empty((len(rows), width + 1), dtype=np.int64)
        out[:, 0] = self.cls_idx
        out[:, 1:] = body
        return out

    def encode_a2m_records(self, sequences: Sequence[str]) -> np.ndarray:
        """Raw a2m record strings -> int64 [R, L+1] (clean-up + tokenisation)."""
        rows = []
        for s in sequences:
            ids = self._a2m_lut[np.frombuffer(s.encode("latin-1", "replace"), dtype=np.uint8)]
            rows.append(ids[ids != _DROP])
        return self._finish(rows)

    def encode(self, sequences: Sequence[str]) -> np.ndarray:
        """Already-clean aligned sequences (characters from AGCUXN-) -> int64 [R, L+1]."""
        if isinstance(sequences, str):
            sequences = [sequences]
        rows = [self._clean_lut[np.frombuffer(s.encode("latin-1", "replace"), dtype=np.uint8)] for s in sequences]
        return self._finish(rows)
