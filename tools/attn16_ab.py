#!/usr/bin/env python3
"""Timing of the 16-bit attention kernels at cfg3 (R=256, C=512, H=12) in the three operand modes.
VARIANTS=1,2,.. sweeps rnamsm_set_param("attn16", v): 1 default, 2 = 128x128 row kernels, 4 = column attention with one query
block per wave, 5 = column attention on the TRACKED loop only.  The column kernel is timed with plane outputs, as the forward runs it."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import _lib, ops
R, C, H = int(os.environ.get("R", 256)), int(os.environ.get("C", 512)), 12
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "1").split(",")]
D = 64 * H
dev = torch.device("cuda:0")
torch.manual_seed(0)
qkv = torch.randn(R * C, 3 * D, device=dev)
def timeit(fn, n=7):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[2:])
lib = _lib.load()
MODES = os.environ.get("MODES", "bf16,f16x3").split(",")
for name, split, fmt in (("bf16", 1, 0), ("f16x3", 3, 1)):
    if name not in MODES:
        continue
    hi, lo = ops.split_bf16(qkv, want_lo=split == 3, fmt=fmt)
    v = lambda a, b: (hi[:, a:b], None if lo is None else lo[:, a:b])
    q, k, vv = v(0, D), v(D, 2 * D), v(2 * D, 3 * D)
    part, ns = ops.row_logits16(q, k, R, C, H, fmt=fmt, scale=ops.row_scaling(R))
    probs, pp = ops.softmax_rows_planes(part, split=split, fmt=fmt, plane_scale=4096.0)
    line = [name]
    for rnd in range(int(os.environ.get("ROUNDS", 1))):
        for var in VARIANTS:
            for _ in (0,):
                for _ in (0,):
                    _lib.check(lib.rnamsm_set_param(b"attn16", var))
                    t1 = timeit(lambda: ops.row_logits16(q, k, R, C, H, fmt=fmt, scale=ops.row_scaling(R)))
                    t2 = timeit(lambda: ops.row_apply16(pp, vv, R, C, H, fmt=fmt, out_scale=1 / 4096.0))
                    t3 = timeit(lambda: ops.col_attn16(q, k, vv, R, C, H, fmt=fmt, scale=0.125, out_planes=True))
                    t4 = timeit(lambda: ops.col_attn16(q, k, vv, R, C, H, fmt=0, out_planes=True, prescaled=True)) if fmt == 0 else float("nan")
                    line.append(f"[v{var}] logits {t1:.3f} apply {t2:.3f} col {t3:.3f} col(prescaled q) {t4:.3f} ms")
    _lib.check(lib.rnamsm_set_param(b"attn16", 1))
    print(" ".join(line), flush=True)
