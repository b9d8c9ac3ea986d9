#!/usr/bin/env python3
"""Go / no-go numerics of folding LayerNorm into the GEMM that consumes it.

    y = LN(x) W^T + b  =  rstd * ( x (W*gamma)^T  -  mean * c ) + d,     c[n] = sum_k (W*gamma)[n,k],  d = b + W beta

so the GEMM can read the residual stream x itself (no normalised copy is ever written) and the per-row statistics are
applied to the accumulators.  Emulated here on the CPU inside the oracle (layer_norm / linear patched), 10 layers,
against the oracle in float64:

    ref32      the oracle in fp32 (the reference's arithmetic)
    fold32     folded form in fp32, exact two-pass statistics
    fold32ssq  folded form in fp32, statistics from fp32 partial (sum, sum of squares) over 64-column slabs
    bf16       every Linear with bf16-rounded operands, fp32 accumulation (what the bf16 mode does today)
    foldbf16   the same with the three LN-fed Linears reading bf16(x) and bf16(W*gamma)
    f16x3 / foldf16x3   operands carried as fp16 hi+lo pairs (22 bits)

Usage: python tests/analysis/ln_fold_numerics.py [M L]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "rna-msm_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch

import truth
from oracle import msm_oracle as O
from rnamsm import synthetic

M, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 128)
torch.set_grad_enabled(False)
EPS = O.LN_EPS

orig_ln, orig_linear = O.layer_norm, O.linear


def rnd(t, mode):
    if mode == "bf16":
        return t.bfloat16().float()
    if mode == "f16x3":
        hi = t.half()
        lo = (t - hi.float()).half()
        return hi.float() + lo.float()
    return t


class Fold:
    def __init__(self, op_round, fold, ssq=False):
        self.op_round, self.fold, self.ssq = op_round, fold, ssq
        self.pending = None

    def layer_norm(self, x, gamma, beta):
        if not self.fold or x.dtype == torch.float64 or self.keep_ln(gamma):
            return orig_ln(x, gamma, beta)
        if self.ssq:
            xs = x.reshape(*x.shape[:-1], -1, 64)
            s1 = xs.sum(-1).sum(-1, keepdim=True)
            s2 = (xs * xs).sum(-1).sum(-1, keepdim=True)
            mu = s1 / x.shape[-1]
            var = s2 / x.shape[-1] - mu * mu
        else:
            mu = x.mean(-1, keepdim=True)
            var = ((x - mu) ** 2).mean(-1, keepdim=True)
        self.pending = (x, mu, torch.rsqrt(var + EPS), gamma, beta)
        return x

    def keep_ln(self, gamma):
        return any(gamma is self.params[k] for k in ("emb_layer_norm_before.weight", "emb_layer_norm_after.weight"))

    def linear(self, x, w, b):
        if x.dtype == torch.float64:
            return orig_linear(x, w, b)
        if self.pending is not None and x is self.pending[0]:
            _, mu, rstd, gamma, beta = self.pending
            wg = rnd(w * gamma, self.op_round)
            c = wg.double().sum(1).float()
            d = (b.double() + w.double() @ beta.double()).float()
            acc = rnd(x, self.op_round) @ wg.t()
            return rstd * (acc - mu * c) + d
        return rnd(x, self.op_round) @ rnd(w, self.op_round).t() + b


def run(tokens, mode):
    op_round = "bf16" if "bf16" in mode else ("f16x3" if "f16x3" in mode else None)
    f = Fold(op_round, mode.startswith("fold"), mode.endswith("ssq"))
    f.params = truth.params(torch.float32, "cpu")
    O.layer_norm, O.linear = f.layer_norm, f.linear
    try:
        res = O.forward(torch.from_numpy(tokens), f.params)
        emb, atp = O.pack_outputs(res)
    finally:
        O.layer_norm, O.linear = orig_ln, orig_linear
    return emb.double(), atp.double()


if __name__ == "__main__":
    toks = synthetic.make_tokens(M, L, 0)
    t_emb, t_atp = truth.oracle_outputs(toks, torch.float64, "cpu")
    print(f"M={M} L={L}")
    for mode in ("ref32", "fold32", "fold32ssq", "f16x3", "foldf16x3", "bf16", "foldbf16"):
        e = truth.errors(*run(toks, mode), t_emb, t_atp)
        print(f"  {mode:10s} emb rel-L2 {e['emb_rel_l2']:.3e}  atp rel-L2 {e['atp_rel_l2']:.3e}  atp max {e['atp_max_abs']:.3e}"
              f"  atp mean {e['atp_mean_abs']:.3e}", flush=True)
