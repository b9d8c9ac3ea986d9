#!/usr/bin/env python3
"""Plane-input 16-bit GEMMs at the cfg3 shapes (T = 131072), as rnamsm_forward runs them: QKV / fc1 with plane outputs,
out_proj / fc2 with the fp32 residual.  ms per launch and executed TFLOP/s per mode.
    python tools/gemm16_planes_bench.py [bf16 f16x3]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import ops
from rnamsm._lib import ACT_GELU_ERF, ACT_NONE
T = int(os.environ.get("T", 131072))
dev = torch.device("cuda:0")
torch.manual_seed(0)
modes = {"bf16": (1, 0), "f16x3": (3, 1)}
if "MFMA16" in os.environ:
    ops.set_param("gemm16_mfma16", int(os.environ["MFMA16"]))      # 0 / 1 / 2: which GEMMs take the 16x16x32 kernel


def timed(fn, reps=5, rounds=7):
    ts = []
    for r in range(rounds):
        torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        if r: ts.append(a.elapsed_time(b) / reps)
    return statistics.median(ts)


for name in (sys.argv[1:] or ["bf16", "f16x3"]):
    split, fmt = modes[name]
    lo = split == 3
    total = 0.0
    for tag, N, K, act, res, planes_out, per_layer in (("qkv", 2304, 768, ACT_NONE, False, True, 2), ("out", 768, 768, ACT_NONE, True, False, 2),
                                                       ("fc1", 3072, 768, ACT_GELU_ERF, False, True, 1), ("fc2", 768, 3072, ACT_NONE, True, False, 1)):
        a = ops.split_bf16(torch.randn(T, K, device=dev), want_lo=lo, fmt=fmt)
        w = ops.split_bf16(torch.randn(N, K, device=dev) * 0.04, want_lo=lo, fmt=fmt)
        b = torch.randn(N, device=dev) * 0.05
        r = torch.randn(T, N, device=dev) if res else None
        fn = (lambda: ops.linear_planes(a, w, b, act=act, out_planes=True, fmt=fmt)) if planes_out else (lambda: ops.linear_planes(a, w, b, residual=r, fmt=fmt))
        t = timed(fn)
        total += per_layer * t
        fl = 2.0 * T * N * K
        print(f"{name:6s} {tag:4s} {t:.4f} ms  {fl / t / 1e9:.0f} TFLOP/s algorithmic ({fl / t / 1e9 / 2500:.3f} of 2.5 PF; x{3 if lo else 1} executed)")
    print(f"{name:6s} six GEMMs of a layer: {total:.3f} ms = {2.474e12 * T / 131072 / total / 1e9:.0f} TFLOP/s")
