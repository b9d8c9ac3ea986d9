#!/usr/bin/env python3
"""Which kernel sets the exact path's distance from the fp64 truth?  (r02: at M=256 x L=512 the HIP fp32 forward is 2.2x
the CPU oracle's fp32 error, at the other BASELINE sizes it is equal.)  Runs the layer-wise path with one knob changed at a
time -- here the length of the fp32 accumulation chains of the tied row logits (slab count of rnamsm_row_logits_chunked) --
and prints each variant's error against the oracle evaluated in fp64 on the device.  Diagnostic only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "rna-msm_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import truth
from rnamsm import ops, synthetic
from rnamsm.model import MSATransformer

M, L = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (256, 512)))
dev = "cuda:0"
toks = synthetic.make_tokens(M, L, 0)
m = MSATransformer(num_layers=10)
m.load_state_dict({k: torch.from_numpy(v) for k, v in truth.state().items()}, strict=True)
m = m.eval().to(dev)
t_emb, t_atp = truth.oracle_outputs(toks, torch.float64, dev)
t = torch.from_numpy(toks).to(dev)


def run(label):
    res = m(t[None], repr_layers=[0, 10], need_head_weights=True)          # layer-wise path (module by module)
    emb = res["representations"][10][0, 0, 1:]
    atp = res["row_attentions"][0][..., 1:, 1:].reshape(-1, L - 1, L - 1)
    e = truth.errors(emb, atp, t_emb, t_atp)
    print(json.dumps({label: {k: float(f"{v:.3e}") for k, v in e.items()}}), flush=True)


run("default")
orig = ops.row_logits
for rpc in (64, 16, 8, 4, 2):
    ops.row_logits = lambda q, k, R, C, H, rows_per_chunk=0, _r=rpc: orig(q, k, R, C, H, rows_per_chunk=_r)
    run(f"row_logits slabs of {rpc} rows (chains of {rpc * 64})")
ops.row_logits = orig
