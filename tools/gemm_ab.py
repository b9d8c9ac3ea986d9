#!/usr/bin/env python3
"""In-process interleaved A/B timing of the Linear GEMM shapes of the forward (cfg3: T = 131072) under tuning knobs.
usage: python tools/gemm_ab.py name=v1,v2,...   e.g.  gemm_group=1,8  or  T=8192 python tools/gemm_ab.py gemm_tile=1,2,0"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import _lib, ops
from rnamsm._lib import ACT_GELU_ERF, ACT_NONE

name, vals = sys.argv[1].split("=")
vals = [int(v) for v in vals.split(",")]
T = int(os.environ.get("T", 131072))
lib = _lib.load()
dev = torch.device("cuda:0")
shapes = [("qkv", 2304, 768, ACT_NONE, False), ("out", 768, 768, ACT_NONE, True),
          ("fc1", 3072, 768, ACT_GELU_ERF, False), ("fc2", 768, 3072, ACT_NONE, True)]
for tag, N, K, act, res in shapes:
    a = torch.randn(T, K, device=dev); w = torch.randn(N, K, device=dev) * 0.04; b = torch.randn(N, device=dev)
    out = torch.empty(T, N, device=dev); r = torch.randn(T, N, device=dev) if res else None
    times = {v: [] for v in vals}
    for rep in range(7):
        for v in vals:
            lib.rnamsm_set_param(name.encode(), v)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.linear(a, w, b, act=act, residual=r, out=out)
            e1.record(); torch.cuda.synchronize()
            if rep > 0:
                times[v].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * T * N * K
    print(tag, " ".join(f"{name}={v}: {statistics.median(t):.3f} ms {fl / statistics.median(t) / 1e9:.1f} TF (min {min(t):.3f})" for v, t in times.items()))
