#!/usr/bin/env python3
"""16-bit modes on mid-size alignments: the persistent 256x256 GEMM walk with a padding-free XCD group (gemm_group = 0, the
default: xcd_group_for_persistent) against the fixed groups of 8 row panels it replaces (gemm_group = 8), one process."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import _lib, synthetic
from rnamsm.model import MSATransformer
dev = torch.device("cuda:0"); lib = _lib.load()
state = synthetic.make_state_dict(seed=0)
m = MSATransformer(num_layers=10)
m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
m = m.eval().to(dev); m.check_finite = False


def timeit(fn, n=8):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[2:])


for (M, L) in ((64, 128), (512, 36), (100, 100), (72, 256), (128, 256), (256, 512)):
    toks = torch.from_numpy(synthetic.make_tokens(M, L, 0)).to(dev)
    line = f"M={M} L={L} ({M * L} tokens, {(M * L + 255) // 256} row panels):"
    for mode in ("bf16", "f16x3"):
        m.gemm_dtype = mode
        res = {}
        for rnd in range(2):
            for g in (0, 8):
                _lib.check(lib.rnamsm_set_param(b"gemm_group", g))
                m.forward_one(toks); torch.cuda.synchronize()
                res.setdefault(g, []).append(timeit(lambda: m.forward_one(toks)))
        a, b = min(res[0]), min(res[8])
        line += f" | {mode}: default {a:.2f} ms, groups of 8 {b:.2f} ms, x{b / a:.3f}"
    _lib.check(lib.rnamsm_set_param(b"gemm_group", 0))
    print(line, flush=True)
