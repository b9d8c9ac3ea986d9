#!/usr/bin/env python3
"""Runs the four plain-bf16 plane GEMMs of a layer at T = 131072 a few times with the 256x256 kernels (gemm16_pp=0) and with
the epilogue-hiding kernel (gemm16_pp=1): the workload of tools/prof_gemm16_pp.sh (rocprofv3 kernel trace + SQ counters)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import _lib, ops
from rnamsm._lib import ACT_GELU_ERF, ACT_NONE
dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(0)
T = 131072
for tag, N, K, act, res, opl in [("qkv", 2304, 768, ACT_NONE, False, True), ("out", 768, 768, ACT_NONE, True, False),
                                 ("fc1", 3072, 768, ACT_GELU_ERF, False, True), ("fc2", 768, 3072, ACT_NONE, True, False)]:
    a = torch.randn(T, K, device=dev); w = torch.randn(N, K, device=dev) * 0.04; b = torch.randn(N, device=dev) * 0.05
    r = torch.randn(T, N, device=dev) if res else None
    ap = ops.split_bf16(a, want_lo=False); wp = ops.split_bf16(w, want_lo=False)
    out = None if opl else torch.empty(T, N, device=dev)
    for v in (0, 1):
        _lib.check(lib.rnamsm_set_param(b"gemm16_pp", v))
        for _ in range(4):
            ops.linear_planes(ap, wp, b, act=act, residual=r, out=out, out_planes=opl)
        torch.cuda.synchronize()
_lib.check(lib.rnamsm_set_param(b"gemm16_pp", 0))
