#!/usr/bin/env python3
"""Timing of the attention kernels at cfg3 (R=256, C=512, H=12) + accuracy vs fp64 on a small column subset."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import _lib, ops
R, C, H = int(os.environ.get("R", 256)), int(os.environ.get("C", 512)), 12
D = 64 * H
dev = torch.device("cuda:0")
torch.manual_seed(0)
qkv = torch.randn(R * C, 3 * D, device=dev)
qkv[:, :D] *= 0.125 * 1.5
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
def timeit(fn, n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[1:])
ctx = torch.empty(R * C, D, device=dev)
t = timeit(lambda: ops.col_attn(q, k, v, R, C, H, out=ctx))
print(f"col_attn   {t:.3f} ms  {4.0 * C * H * R * R * 64 / t / 1e9:.1f} TF")
# accuracy on 4 columns, all heads
cs = [0, 1, C // 2, C - 1]
q64 = q.view(R, C, H, 64)[:, cs].double(); k64 = k.view(R, C, H, 64)[:, cs].double(); v64 = v.view(R, C, H, 64)[:, cs].double()
p = torch.softmax(torch.einsum("ichd,jchd->hcij", q64, k64), -1)
want = torch.einsum("hcij,jchd->ichd", p, v64)
got = ctx.view(R, C, H, 64)[:, cs].double()
print("col_attn rel-L2 vs fp64:", float((got - want).norm() / want.norm()))
qkv[:, :D] *= (1.0 / 16)
part, ns = ops.row_logits(q, k, R, C, H)
t = timeit(lambda: ops.row_logits(q, k, R, C, H))
print(f"row_logits {t:.3f} ms  {2.0 * H * C * C * R * 64 / t / 1e9:.1f} TF (nsplit {ns})")
probs = ops.softmax_rows(part)
t = timeit(lambda: ops.row_apply(probs, v, R, C, H, out=ctx))
print(f"row_apply  {t:.3f} ms  {2.0 * H * C * C * R * 64 / t / 1e9:.1f} TF")
