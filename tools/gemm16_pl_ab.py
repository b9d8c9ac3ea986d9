#!/usr/bin/env python3
"""Throughput of the plane-input 16-bit GEMMs (LDS-DMA kernels) at the cfg3 shapes: bf16 (split 1) and f16x3.
ONLY=<tag> restricts to one shape (for PMC runs)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import ops
from rnamsm._lib import ACT_GELU_ERF, ACT_NONE
T = int(os.environ.get("T", 131072))
ONLY = os.environ.get("ONLY")
REPS = int(os.environ.get("REPS", 6))
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "2,3").split(",")]
KNOB = os.environ.get("KNOB", "gemm16_dma").encode()      # which rnamsm_set_param knob VARIANTS sweeps
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timeit(fn, n=REPS):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[min(2, n - 1):])
for tag, N, K, act, res, opl in [("qkv", 2304, 768, ACT_NONE, False, True), ("out", 768, 768, ACT_NONE, True, False),
                                 ("fc1", 3072, 768, ACT_GELU_ERF, False, True), ("fc2", 768, 3072, ACT_NONE, True, False)]:
    if ONLY and tag != ONLY: continue
    a = torch.randn(T, K, device=dev); w = torch.randn(N, K, device=dev) * 0.04; b = torch.randn(N, device=dev) * 0.05
    r = torch.randn(T, N, device=dev) if res else None
    out = None if opl else torch.empty(T, N, device=dev)
    line = [tag]
    fl = 2.0 * T * N * K
    for name, split, fmt in (("bf16", 1, 0), ("f16x3", 3, 1)):
        ap = ops.split_bf16(a, want_lo=split == 3, fmt=fmt)
        wp = ops.split_bf16(w, want_lo=split == 3, fmt=fmt)
        fn = lambda: ops.linear_planes(ap, wp, b, act=act, residual=r, out=out, out_planes=opl, fmt=fmt)
        from rnamsm import _lib
        ts = []
        for variant in VARIANTS:
            _lib.check(_lib.load().rnamsm_set_param(KNOB, variant))
            fn(); torch.cuda.synchronize()
            ts.append(timeit(fn))
        line.append(f"{name}: " + " / ".join(f"v{v} {t:.3f} ms {fl / t / 1e9:.0f} TF" for v, t in zip(VARIANTS, ts)))
    print(" | ".join(line), flush=True)
