#!/usr/bin/env python3
"""Does the row stride of the operand planes matter (L2 / HBM channel aliasing)?  The plain-bf16 plane GEMM at K and K +- 64
(row strides that are / are not multiples of 2 KB), same M and N: TF/s should be flat in K if the stride is harmless."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import _lib, ops
from rnamsm._lib import ACT_NONE
dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(0)


def timeit(fn, n=10):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[2:])


for (M, N, Ks, res) in [(4096, 4096, (3968, 4032, 4096, 4160), False), (8192, 8192, (8128, 8192), False),
                        (131072, 768, (2944, 3008, 3072, 3136, 3200), True), (131072, 2304, (704, 768, 832), False),
                        (131072, 3072, (704, 768, 832), False)]:
    line = f"M={M} N={N}:"
    for K in Ks:
        a = torch.rand(M, K, device=dev) * 2 - 1; w = torch.rand(N, K, device=dev) * 2 - 1; b = torch.zeros(N, device=dev)
        ap = ops.split_bf16(a, want_lo=False); wp = ops.split_bf16(w, want_lo=False)
        del a, w
        r = torch.randn(M, N, device=dev) if res else None
        out = torch.empty(M, N, device=dev) if res else None
        fn = lambda: ops.linear_planes(ap, wp, b, act=ACT_NONE, residual=r, out=out, out_planes=not res)
        fn(); t = timeit(fn)
        line += f" | K={K} (stride {2 * K} B) {t:.3f} ms {2.0 * M * N * K / t / 1e9:5.0f} TF"
        del ap, wp, r, out
        torch.cuda.empty_cache()
    print(line, flush=True)
