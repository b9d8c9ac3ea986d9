#!/usr/bin/env python3
"""A/B of the two fp32 column-attention kernels in one process (knob "col_dma": 1 = LDS-DMA staging, 32-key chunks, three
blocks per CU; 0 = register-staged, 64-key chunks, two blocks per CU): interleaved rounds, bit-identical outputs required.
Inputs are a LayerNorm-like activation through a random QKV projection (the magnitudes the forward produces)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import _lib, ops
H = 12
D = 64 * H
dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(0)


def timeit(fn, n=7):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[2:]), min(ts[2:])


for R, C in ((256, 512), (128, 256), (64, 128), (512, 36), (1024, 1024), (100, 300)):
    qkv = torch.randn(R * C, 3 * D, device=dev)
    qkv[:, :D] *= 0.125 * 1.5
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    outs = {}
    res = {0: [], 1: []}
    for rnd in range(3):
        for knob in (0, 1):
            _lib.check(lib.rnamsm_set_param(b"col_dma", knob))
            ctx = torch.empty(R * C, D, device=dev)
            med, mn = timeit(lambda: ops.col_attn(q, k, v, R, C, H, out=ctx))
            res[knob].append(med)
            outs[knob] = ctx
    same = torch.equal(outs[0], outs[1])
    fl = 4.0 * C * H * R * R * 64
    a, b = statistics.median(res[0]), statistics.median(res[1])
    print(f"R={R:4d} C={C:4d}  regs-staged {a:.3f} ms ({fl / a / 1e9:6.1f} TF)   lds-dma {b:.3f} ms ({fl / b / 1e9:6.1f} TF)   "
          f"ratio {a / b:.3f}   bit-identical {same}", flush=True)
    # padded variant
    mask = (torch.rand(R * C, device=dev) < 0.1).to(torch.uint8)
    for knob in (0, 1):
        _lib.check(lib.rnamsm_set_param(b"col_dma", knob))
        ctx = torch.empty(R * C, D, device=dev)
        med, mn = timeit(lambda: ops.col_attn(q, k, v, R, C, H, out=ctx, pad_mask=mask))
        outs[knob] = ctx
        res[knob] = med
    print(f"        masked: regs-staged {res[0]:.3f} ms   lds-dma {res[1]:.3f} ms   bit-identical {torch.equal(outs[0], outs[1])}", flush=True)
_lib.check(lib.rnamsm_set_param(b"col_dma", -1))
