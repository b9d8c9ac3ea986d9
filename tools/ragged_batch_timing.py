#!/usr/bin/env python3
"""Alignments of different shapes one by one against one forward_ragged call (padded frame, true depths): ms per MSA."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import numpy as np
import torch
from rnamsm import synthetic
from rnamsm.model import MSATransformer
dev = "cuda:0"
model = MSATransformer(num_layers=10)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(seed=0).items()}, strict=True)
model = model.eval().to(dev)
rng = np.random.default_rng(0)
for (rlo, rhi), (clo, chi), n in (((2, 12), (40, 80), 16), ((8, 24), (60, 120), 16), ((16, 48), (80, 160), 8), ((40, 64), (100, 140), 8)):
    shapes = [(int(rng.integers(rlo, rhi + 1)), int(rng.integers(clo, chi + 1))) for _ in range(n)]
    msas = [torch.from_numpy(synthetic.make_tokens(r, c, i)).to(dev) for i, (r, c) in enumerate(shapes)]
    real = sum(r * c for r, c in shapes)
    frame = n * max(r for r, _ in shapes) * max(c for _, c in shapes)
    for _ in range(2):
        model.forward_ragged(msas); [model.forward_one(t, has_padding=False, need_repr=False) for t in msas]
    torch.cuda.synchronize()
    b1 = b2 = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for t in msas:
            model.forward_one(t, has_padding=False, need_repr=False)
        torch.cuda.synchronize(); b1 = min(b1, time.perf_counter() - t0)
        t0 = time.perf_counter()
        model.forward_ragged(msas)
        torch.cuda.synchronize(); b2 = min(b2, time.perf_counter() - t0)
    print(f"{n} MSAs, rows {rlo}..{rhi}, columns {clo}..{chi} ({real} real tokens in a frame of {frame}): one by one {1e3 * b1 / n:.3f} ms/MSA, "
          f"ragged batch {1e3 * b2 / n:.3f} ms/MSA (x{b1 / b2:.2f}), {real / b2:.0f} real residues/s", flush=True)
