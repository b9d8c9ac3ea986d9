#!/usr/bin/env python3
"""The plain-bf16 plane GEMM (plane in, plane out) at square and tall shapes under the block-order / persistence knobs, one process:
how far the persistent walk is from an even distribution of tiles when there are few row panels (gemm_group = 8 on 16 panels: a
quarter of the blocks owned every tile, 432 TF at 4096^3; xcd_group_for_persistent: 1156 TF), and where the K loop stands at long K
(1270-1290 TF) against the 1320-1340 TF cdna_hip_programming.md quotes for its 8-phase template at 4096^3."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import _lib, ops
from rnamsm._lib import ACT_NONE
dev = torch.device("cuda:0"); lib = _lib.load(); torch.manual_seed(0)
def timeit(fn, n=10):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[2:])
def setp(**kw):
    for k, v in kw.items(): _lib.check(lib.rnamsm_set_param(k.encode(), v))
for (M, N, K) in [(4096, 4096, 4096), (4096, 4096, 1024), (8192, 4096, 4096), (16384, 4096, 4096), (65536, 4096, 4096), (4096, 2304, 4096), (36864, 2304, 4096)]:
    a = torch.rand(M, K, device=dev) * 2 - 1; w = torch.rand(N, K, device=dev) * 2 - 1; b = torch.zeros(N, device=dev)
    ap = ops.split_bf16(a, want_lo=False); wp = ops.split_bf16(w, want_lo=False); del a, w
    fn = lambda: ops.linear_planes(ap, wp, b, act=ACT_NONE, residual=None, out=None, out_planes=True)
    line = f"M={M} N={N} K={K} tiles/CU {M*N/65536/256:.1f}:"
    for name, kw in (("default", {}), ("swp", dict(gemm16_mfma16=0)), ("swp stagger 16k", dict(gemm16_mfma16=0, gemm16_stagger=16000)),
                     ("swp stagger 64k", dict(gemm16_mfma16=0, gemm16_stagger=64000)), ("swp group1", dict(gemm16_mfma16=0, gemm_group=1)),
                     ("swp group2", dict(gemm16_mfma16=0, gemm_group=2)), ("swp group16", dict(gemm16_mfma16=0, gemm_group=16)),
                     ("swp nonpersist", dict(gemm16_mfma16=0, gemm16_persist=0))):
        setp(gemm16_mfma16=1, gemm16_stagger=0, gemm_group=0, gemm16_persist=256); setp(**kw)
        fn(); t = timeit(fn)
        line += f" | {name} {t:.3f} ms {2.0*M*N*K/t/1e9:.0f} TF"
    setp(gemm16_mfma16=1, gemm16_stagger=0, gemm_group=0, gemm16_persist=256)
    print(line, flush=True)
    del ap, wp; torch.cuda.empty_cache()
