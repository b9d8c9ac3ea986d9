#!/usr/bin/env python3
"""Authoring-container-only fuzz: random a2m texts (every character class, lowercase / '.' / '*' insertions, multi-line
records, blank lines, CRLF, ragged lengths, invalid characters) through the imported reference (MSA.from_fasta +
Vocab.encode, utils/align.py:291-317, utils/tokenization.py:107-165) and through rnamsm.msa + rnamsm.alphabet: the int64
tokens must be bit-identical, or both sides must raise the same exception type; alignments that parse are also
sub-sampled by both (`select_diverse` diversity-max / -min vs rnamsm.msa.greedy_select: index-exact, ties included).  Needs /root/reference (see _refimport);
not collected by pytest, never runs on the GPU box.

    python tests/golden/fuzz_tokens_vs_reference.py [cases [seed]]
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))

import numpy as np

import _refimport
from rnamsm.alphabet import RNAAlphabet
from rnamsm.msa import greedy_select, load_msa_tokens

ref_model, ref_modules, ref_msm, Vocab, MSA = _refimport.reference_modules()
vocab = Vocab.from_esm_alphabet(ref_msm.data.Alphabet.from_architecture("rna language"))
alphabet = RNAAlphabet.from_architecture("rna language")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
VALID = "ACGUTXN-RYKMSWBDHV"
INSERT = "acgutnxrykmswbdhv.*"
INVALID = "EFIJLOPQZ0123 _?#\t"


def make_text():
    depth, width = int(rng.integers(1, 12)), int(rng.integers(0, 40))
    kind = rng.random()
    lines = []
    for r in range(depth):
        w = width if (kind > 0.1 or r == 0) else max(0, width + int(rng.integers(-2, 3)))      # 10 %: ragged
        seq = []
        for _ in range(w):
            seq.append(VALID[int(rng.integers(len(VALID)))])
            while rng.random() < 0.15:
                seq.append(INSERT[int(rng.integers(len(INSERT)))])
        if rng.random() < 0.05 and seq:                                                        # 5 % of records: an invalid character
            seq[int(rng.integers(len(seq)))] = INVALID[int(rng.integers(len(INVALID)))]
        s = "".join(seq)
        lines.append(f">seq{r} some description/{r}")
        if len(s) > 6 and rng.random() < 0.3:                                                  # multi-line record
            cut = int(rng.integers(1, len(s)))
            lines += [s[:cut], s[cut:]]
        else:
            lines.append(s)
        if rng.random() < 0.1:
            lines.append("")
    eol = "\r\n" if rng.random() < 0.15 else "\n"
    return eol.join(lines) + (eol if rng.random() < 0.8 else "")


def outcome(fn):
    try:
        return "ok", fn()
    except (AssertionError, ValueError, IndexError, KeyError) as e:
        return type(e).__name__, None


mismatch = 0
counts = {}
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "x.a2m_msa2")
    for case in range(cases):
        text = make_text()
        with open(path, "w", newline="") as f:
            f.write(text)
        ref_kind, ref_tok = outcome(lambda: vocab.encode(MSA.from_fasta(path)))
        our_kind, our_tok = outcome(lambda: load_msa_tokens(path, alphabet, max_seqs_per_msa=None))
        counts[ref_kind] = counts.get(ref_kind, 0) + 1
        same = ref_kind == our_kind and (ref_tok is None or (ref_tok.dtype == our_tok.dtype and ref_tok.shape == our_tok.shape
                                                             and np.array_equal(ref_tok, our_tok)))
        if same and ref_tok is not None and ref_tok.shape[0] >= 3 and ref_tok.shape[1] > 1:
            # greedy max / min-Hamming sub-sampling (utils/align.py:128-148) of the same alignment: many ties at these sizes
            n = int(rng.integers(2, ref_tok.shape[0]))
            for mode in ("max", "min"):
                want = vocab.encode(MSA.from_fasta(path).select_diverse(n, method=f"diversity-{mode}"))
                got = our_tok[greedy_select(our_tok, n, mode)]
                if not np.array_equal(want, got):
                    same = False
                    print(f"sub-sampling diversity-{mode} n={n} differs")
            counts["subsampled"] = counts.get("subsampled", 0) + 1
        if not same:
            mismatch += 1
            print(f"MISMATCH case {case}: reference {ref_kind} {None if ref_tok is None else ref_tok.shape}, "
                  f"ours {our_kind} {None if our_tok is None else our_tok.shape}\n{text!r}")
print(f"{cases} a2m texts: reference outcomes {counts}; {mismatch} mismatches")
sys.exit(1 if mismatch else 0)
