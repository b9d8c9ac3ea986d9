#!/usr/bin/env python3
"""B same-shape small MSAs one by one (rnamsm_forward each) against one rnamsm_forward_batch call: ms per MSA, residues/s.
    python tools/batch_small_msas.py"""
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
for M, L in ((4, 64), (8, 64), (16, 64), (16, 128), (32, 128), (64, 128), (128, 128)):
    line = []
    for B in (2, 4, 8, 16, 32):
        if B * M * L > 65536:
            continue
        toks = torch.from_numpy(np.stack([synthetic.make_tokens(M, L, b) for b in range(B)])).to(dev)
        for _ in range(2):
            model.forward_batch(toks); [model.forward_one(toks[b], has_padding=False) for b in range(B)]
        torch.cuda.synchronize()
        best_one = best_bat = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for b in range(B):
                model.forward_one(toks[b], has_padding=False)
            torch.cuda.synchronize()
            best_one = min(best_one, (time.perf_counter() - t0) / B)
            t0 = time.perf_counter()
            model.forward_batch(toks)
            torch.cuda.synchronize()
            best_bat = min(best_bat, (time.perf_counter() - t0) / B)
        line.append(f"B={B}: {1e3 * best_bat:.3f} ms/MSA (x{best_one / best_bat:.2f}, {M * L / best_bat:7.0f} res/s)")
    print(f"M={M:3d} L={L:3d} one by one {1e3 * best_one:.3f} ms/MSA | batched " + "  ".join(line), flush=True)
