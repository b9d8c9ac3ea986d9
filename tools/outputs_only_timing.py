#!/usr/bin/env python3
"""Full forward vs the outputs-only forward (need_repr=False: what the CLI asks for) at the BASELINE shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import synthetic
from rnamsm.model import MSATransformer
m = MSATransformer(num_layers=10)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(seed=0).items()}, strict=True)
m = m.eval().cuda()
for (M, L) in ((256, 512), (128, 256), (64, 128), (512, 36), (1024, 1024)):
    t = torch.from_numpy(synthetic.make_tokens(M, L, 0)).cuda()
    res = {}
    for need in (True, False):
        m.forward_one(t, has_padding=False, need_repr=need); torch.cuda.synchronize()
        n = 6 if M < 1024 else 2
        t0 = time.perf_counter()
        for _ in range(n):
            m.forward_one(t, has_padding=False, need_repr=need)
        torch.cuda.synchronize()
        res[need] = (time.perf_counter() - t0) / n
    print(f"M={M} L={L}: full {1e3 * res[True]:.2f} ms ({M * L / res[True]:.0f} res/s)   outputs-only {1e3 * res[False]:.2f} ms "
          f"({M * L / res[False]:.0f} res/s)   x{res[True] / res[False]:.3f}", flush=True)
