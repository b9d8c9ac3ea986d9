#!/usr/bin/env python3
"""Latency of ONE fp32 GEMM launch at tiny / small M (a lone small alignment's GEMMs): event pair around a single launch and the
per-launch time inside a queue of 200, for the forward's shapes and some probes (N = 128: one column tile).  Found in round 4: with
padded XCD groups the 18-24 column tiles of a single row panel all landed on 4 CUs (EXPERIMENTS R4.8)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rna-msm_amd'))
import torch
from rnamsm import ops, _lib
dev = 'cuda:0'
def lat(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    # chain of 20 dependent launches (out feeds the next as residual): per-launch time inside a queue
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): fn()
    torch.cuda.synchronize(); q = (time.perf_counter() - t0) / 200 * 1e6
    return statistics.median(ts), q
for M in (82, 520, 2064):
    for N, K in ((2304, 768), (2304, 192), (768, 768), (3072, 768), (768, 3072), (128, 768), (128, 32)):
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.04; b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev)
        l, q = lat(lambda: ops.linear(a, w, b, out=out))
        print(f"M={M} N={N} K={K}: single launch {l:.1f} us (event pair), in a queue {q:.1f} us per launch")
x = torch.randn(4096, 768, device=dev); g = torch.ones(768, device=dev); bb = torch.zeros(768, device=dev)
l, q = lat(lambda: ops.layernorm(x, g, bb))
print(f"layernorm 4096 rows: single {l:.1f} us, queued {q:.1f} us")
