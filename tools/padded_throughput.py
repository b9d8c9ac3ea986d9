#!/usr/bin/env python3
"""Throughput of a PADDED batch: B ragged MSAs padded to a common M x L run through
MSATransformer.forward with the padding masks active (masked QKV rows, key masks in the row softmax, the masked
column-attention instance), next to the same shapes without padding.  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import numpy as np
import torch
from rnamsm import synthetic
from rnamsm.model import MSATransformer

dev = "cuda:0"
M, L, B = 256, 512, 4
m = MSATransformer(num_layers=10)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(seed=0).items()}, strict=True)
m = m.eval().to(dev)
m.max_tokens_per_msa_(2 ** 30)
full = np.stack([synthetic.make_tokens(M, L, i) for i in range(B)])
padded = full.copy()
real = 0
for b in range(B):                                   # ragged: element b keeps (M - 24 b) rows and (L - 40 b) columns
    r, c = M - 24 * b, L - 40 * b
    padded[b, r:, :] = 1
    padded[b, :, c:] = 1
    real += r * c
out = {"shape": [B, M, L], "real_residues": real, "padded_residues": B * M * L}
for mode in ("f32", "f16x3"):
    m.gemm_dtype = mode
    for name, toks in (("unpadded", full), ("padded", padded)):
        t = torch.from_numpy(toks).to(dev)
        m(t, need_head_weights=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            m(t, need_head_weights=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        out[f"{mode}_{name}"] = {"ms_per_batch": 1e3 * dt, "padded_residues_per_s": B * M * L / dt,
                                 "real_residues_per_s": (real if name == "padded" else B * M * L) / dt}
print(json.dumps(out))
