#!/usr/bin/env python3
"""Per-GEMM cost of the folded LayerNorm at the cfg3 shapes (T = 131072), one process, interleaved:
  producers  out_proj / fc2 + residual      plain  vs  + row partial sums (rnamsm_gemm_residual_stats)
  consumers  QKV / fc1                      LayerNorm output as A  vs  x as A with the fold applied in the epilogue
    T=131072 python tools/ln_fold_gemm_ab.py"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import ops
from rnamsm._lib import ACT_GELU_ERF, ACT_NONE

T = int(os.environ.get("T", 131072))
dev = torch.device("cuda:0")
D, F = 768, 3072


def bench(fns, reps=3, rounds=7):
    times = {k: [] for k in fns}
    for r in range(rounds):
        for k, fn in fns.items():
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record(); torch.cuda.synchronize()
            if r:
                times[k].append(a.elapsed_time(b) / reps)
    return {k: statistics.median(v) for k, v in times.items()}


x = torch.randn(T, D, device=dev)
g, be = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev) * 0.1
for tag, K in (("out_proj", D), ("fc2", F)):
    a = torch.randn(T, K, device=dev) * 0.5
    w, b = torch.randn(D, K, device=dev) * 0.04, torch.randn(D, device=dev)
    out = torch.empty(T, D, device=dev)
    r = bench({"plain": lambda: ops.linear(a, w, b, residual=x, out=out),
               "stats": lambda: ops.linear_residual_stats(a, w, b, x, out=out)})
    print(f"{tag:9s} plain {r['plain']:.4f} ms   + row sums {r['stats']:.4f} ms   ({1e3 * (r['stats'] - r['plain']):+.1f} us)")
xn = ops.layernorm(x, g, be)
part = ops.row_partials(x)
st = ops.row_stats_from_partials(part, D)
for tag, N, act, sc in (("qkv", 3 * D, ACT_NONE, D), ("fc1", F, ACT_GELU_ERF, 0), ("fc1 no act", F, ACT_NONE, 0), ("qkv gelu", 3 * D, ACT_GELU_ERF, 0)):
    w, b = torch.randn(N, D, device=dev) * 0.04, torch.randn(N, device=dev)
    wg, c, d = ops.ln_fold_weights(w, b, g, be)
    out = torch.empty(T, N, device=dev)
    r = bench({"plain": lambda: ops.linear(xn, w, b, act=act, scale=0.125, scale_cols=sc, out=out),
               "fold": lambda: ops.linear_lnfold(x, wg, c, d, st, act=act, scale=0.125, scale_cols=sc, out=out),
               "self": lambda: ops.linear_lnfold(x, wg, c, d, None, act=act, scale=0.125, scale_cols=sc, out=out)})
    print(f"{tag:9s} plain {r['plain']:.4f} ms   folded {r['fold']:.4f} ms ({1e3 * (r['fold'] - r['plain']):+.1f} us)   "
          f"folded, own sums {r['self']:.4f} ms ({1e3 * (r['self'] - r['plain']):+.1f} us)")
ln = bench({"layernorm": lambda: ops.layernorm(x, g, be, out=xn), "row_partials": lambda: ops.row_partials(x),
            "finalize": lambda: ops.row_stats_from_partials(part, D)})
print(f"layernorm {ln['layernorm']:.4f} ms   row_partials {ln['row_partials']:.4f} ms   row_stats_from_partials {ln['finalize']:.4f} ms")
