#!/usr/bin/env python3
"""Accuracy (vs fp64 on a row subset) and throughput of the fp32-MFMA GEMM vs the bf16 / f16x3 GEMM at cfg3 shapes."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import ops
from rnamsm._lib import ACT_GELU_ERF, ACT_NONE
T = int(os.environ.get("T", 131072))
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timeit(fn, n=6):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[2:])
for tag, N, K, act, res in [("qkv", 2304, 768, ACT_NONE, False), ("out", 768, 768, ACT_NONE, True),
                            ("fc1", 3072, 768, ACT_GELU_ERF, False), ("fc2", 768, 3072, ACT_NONE, True)]:
    a = torch.randn(T, K, device=dev); w = torch.randn(N, K, device=dev) * 0.04; b = torch.randn(N, device=dev) * 0.05
    r = torch.randn(T, N, device=dev) if res else None
    hi, lo = ops.split_bf16(w)
    hh, hl = ops.split_bf16(w, fmt=1)
    sub = slice(0, 1024)
    ref = a[sub].double() @ w.double().t() + b.double()
    if act == ACT_GELU_ERF: ref = 0.5 * ref * (1 + torch.erf(ref / 2 ** 0.5))
    if res: ref = ref + r[sub].double()
    out = torch.empty(T, N, device=dev)
    line = [tag]
    fl = 2.0 * T * N * K
    for name, fn in (("f32", lambda: ops.linear(a, w, b, act=act, residual=r, out=out)),
                     ("bf16", lambda: ops.linear_bf16(a, hi, None, b, act=act, residual=r, out=out, split=1, fmt=0)),
                     ("f16x3", lambda: ops.linear_bf16(a, hh, hl, b, act=act, residual=r, out=out, split=3, fmt=1))):
        fn(); torch.cuda.synchronize()
        err = float((out[sub].double() - ref).norm() / ref.norm())
        mx = float((out[sub].double() - ref).abs().max() / ref.abs().max())
        t = timeit(fn)
        line.append(f"{name}: {t:.3f} ms {fl / t / 1e9:.0f} TF rel {err:.1e} max {mx:.1e}")
    print(" | ".join(line))
