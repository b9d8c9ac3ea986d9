#!/usr/bin/env python3
"""256x256 16-bit GEMM (gemm16_swp_kernel): one block per tile vs persistent blocks (one per CU) with a start stagger, at
the M=256 L=512 shapes, interleaved rounds in one process; outputs must be bit-identical.
VARIANTS="persist:stagger,..." (cycles)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import _lib, ops
from rnamsm._lib import ACT_GELU_ERF, ACT_NONE
T = int(os.environ.get("T", 131072))
VARIANTS = [tuple(int(x) for x in v.split(":")) for v in os.environ.get("VARIANTS", "0:0,256:0,256:4000,256:8000,256:16000").split(",")]
ROUNDS = int(os.environ.get("ROUNDS", 3))
dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(0)


def timeit(fn, n=6):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[2:])


def setv(v):
    _lib.check(lib.rnamsm_set_param(b"gemm16_persist", v[0]))
    _lib.check(lib.rnamsm_set_param(b"gemm16_stagger", v[1]))


for tag, N, K, act, res, opl in [("qkv", 2304, 768, ACT_NONE, False, True), ("out", 768, 768, ACT_NONE, True, False),
                                 ("fc1", 3072, 768, ACT_GELU_ERF, False, True), ("fc2", 768, 3072, ACT_NONE, True, False)]:
    a = torch.randn(T, K, device=dev); w = torch.randn(N, K, device=dev) * 0.04; b = torch.randn(N, device=dev) * 0.05
    r = torch.randn(T, N, device=dev) if res else None
    fl = 2.0 * T * N * K
    for name, split, fmt in (("bf16", 1, 0), ("f16x3", 3, 1)):
        ap = ops.split_bf16(a, want_lo=split == 3, fmt=fmt)
        wp = ops.split_bf16(w, want_lo=split == 3, fmt=fmt)
        outs, times = {}, {v: [] for v in VARIANTS}
        for rnd in range(ROUNDS):
            for v in VARIANTS:
                setv(v)
                out = None if opl else torch.empty(T, N, device=dev)
                fn = lambda: ops.linear_planes(ap, wp, b, act=act, residual=r, out=out, out_planes=opl, fmt=fmt)
                res_t = fn(); torch.cuda.synchronize()
                times[v].append(timeit(fn))
                outs[v] = res_t[0] if opl else res_t
        base = outs[VARIANTS[0]]
        same = all(torch.equal(outs[v], base) for v in VARIANTS)
        t0 = statistics.median(times[VARIANTS[0]])
        print(f"{tag:4s} {name:6s} " + " | ".join(f"{v[0]}:{v[1]} {statistics.median(times[v]):.3f} ms {fl / statistics.median(times[v]) / 1e9:5.0f} TF x{t0 / statistics.median(times[v]):.3f}"
                                                 for v in VARIANTS) + f" | identical {same}", flush=True)
setv((0, 0))
