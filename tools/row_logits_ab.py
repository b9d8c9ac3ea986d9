#!/usr/bin/env python3
"""Timing of the exact-path row-attention kernels (K4 logits, K5 softmax) at one shape, with the slab count in use.
    python tools/row_logits_ab.py [R C]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import _lib, ops

R, C = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 512)
H, D = 12, 768
dev = torch.device("cuda:0")
qkv = torch.randn(R * C, 3 * D, device=dev) * 0.3
q, k = qkv[:, :D], qkv[:, D:2 * D]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps)
    return statistics.median(ts)


partial, nsplit = ops.row_logits(q, k, R, C, H)
t4 = timed(lambda: ops.row_logits(q, k, R, C, H))
t5 = timed(lambda: ops.softmax_rows(partial))
fl = 2.0 * H * C * C * R * 64
print(f"R={R} C={C}: {nsplit} slabs; row_logits {t4:.4f} ms = {fl / t4 / 1e9:.1f} TF ({fl / t4 / 1e9 / 157.3:.3f}); softmax_rows {t5:.4f} ms")
