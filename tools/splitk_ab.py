#!/usr/bin/env python3
"""A/B of split-K in the fc2 GEMM of small MSAs (knob "gemm_splitk": 1 = by shape, 0 = off), one process: ms per forward.
Below ~5 k tokens the K = 3072 GEMM has fewer tiles than the chip has block slots (2048 tokens: 96 tiles, 96 K steps each)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import ops, synthetic
from rnamsm.model import MSATransformer
dev = "cuda:0"
KNOBS = (0, 1)
model = MSATransformer(num_layers=10)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(seed=0).items()}, strict=True)
model = model.eval().to(dev)
for M, L in ((4, 64), (8, 64), (16, 64), (32, 64), (16, 128), (21, 128), (32, 128), (42, 128), (64, 128), (512, 36)):
    tok = torch.from_numpy(synthetic.make_tokens(M, L, 0)).to(dev)
    res, outs = {}, {}
    for rnd in range(3):
        for knob in KNOBS:
            ops.set_param("gemm_splitk", knob)
            for _ in range(3): out = model.forward_one(tok, has_padding=False)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            a.record()
            for _ in range(n): out = model.forward_one(tok, has_padding=False)
            b.record(); torch.cuda.synchronize()
            res.setdefault(knob, []).append(a.elapsed_time(b) / n)
            outs[knob] = out
    print(f"M={M:3d} L={L:3d} ({M * L:5d} tokens): " + "  ".join(f"cap {k}: {min(res[k]):.3f} ms" for k in KNOBS), flush=True)
ops.set_param("gemm_splitk", 0)       # the default since round 5 (one arithmetic per alignment)
