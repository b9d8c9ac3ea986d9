#!/usr/bin/env python3
"""Plain-bf16 plane GEMM at square shapes (4096^3, 8192^3; random operands, plane output): where the 256x256 kernels stand
against the known-good reference point of cdna_hip_programming.md 5 (the 256^2 8-phase template: 1320-1340 TF at 4096^3,
~1470 TF at 8192^3 on uniform random operands).  If the loop reaches that, the K = 768 GEMMs of the model are short of it
because of their epilogue / prologue share and HBM traffic, not because of the loop's structure."""
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


for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (131072, 768, 768), (131072, 2304, 768), (131072, 2304, 4096)]:
    a = torch.rand(M, K, device=dev) * 2 - 1; w = torch.rand(N, K, device=dev) * 2 - 1; b = torch.zeros(N, device=dev)
    ap = ops.split_bf16(a, want_lo=False); wp = ops.split_bf16(w, want_lo=False)
    del a, w
    line = f"M={M} N={N} K={K}:"
    for name, knobs in (("16x16x32 (q16)", {"gemm16_mfma16": 2}), ("32x32x16 (swp)", {"gemm16_mfma16": 0})):
        for k, v in knobs.items():
            _lib.check(lib.rnamsm_set_param(k.encode(), v))
        for opl in (True, False):
            out = None if opl else torch.empty(M, N, device=dev)
            fn = lambda: ops.linear_planes(ap, wp, b, act=ACT_NONE, residual=None, out=out, out_planes=opl)
            fn(); t = timeit(fn)
            line += f" | {name} {'bf16' if opl else 'fp32'} out {t:.3f} ms {2.0 * M * N * K / t / 1e9:5.0f} TF"
            del out
    print(line, flush=True)
    _lib.check(lib.rnamsm_set_param(b"gemm16_mfma16", 1))
    del ap, wp
    torch.cuda.empty_cache()
