#!/usr/bin/env python3
"""rnamsm_forward_batch on B alignments of 8 x 64 (exact path) for several B and GEMM knobs, one process: how the residues/s of a
batch of small alignments depends on its token count (GEMM tile quantisation: 576 k at B = 28, 649 k at 32, 603 k at 36, 660 k at 48,
676 k at 64) and that none of gemm_tile / ln_fold / gemm_splitk moves it."""
import os, sys, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import numpy as np, torch
from rnamsm import _lib, synthetic, ops
from rnamsm.model import MSATransformer
dev = torch.device("cuda:0"); lib = _lib.load()
m = MSATransformer(num_layers=10)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(seed=0).items()}, strict=True)
m = m.eval().to(dev)
for (B, R, C) in ((32, 8, 64), (28, 8, 64), (24, 8, 64), (36, 8, 64), (48, 8, 64), (64, 8, 64)):
    st = torch.from_numpy(np.stack([synthetic.make_tokens(R, C, 900 + b) for b in range(B)])).to(dev)
    line = f"B={B} {R}x{C} ({B*R*C} tokens):"
    for knobs in ({}, {"gemm_tile": 1}, {"gemm_tile": 2}, {"ln_fold": 3}, {"ln_fold": 0}, {"gemm_splitk": 1}):
        for k, v in (("gemm_tile", 0), ("ln_fold", 1), ("gemm_splitk", 0)): ops.set_param(k, v)
        for k, v in knobs.items(): ops.set_param(k, v)
        for _ in range(3): m.forward_batch(st, has_padding=False)
        torch.cuda.synchronize(); ts = []
        for _ in range(7):
            t0 = time.perf_counter(); m.forward_batch(st, has_padding=False); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        w = statistics.median(ts)
        line += f" | {knobs or 'default'} {w*1e3:.2f} ms {B*R*C/w/1e3:.0f}k"
    print(line, flush=True)
