#!/usr/bin/env python3
"""A/B of one tuning knob on the whole forward, one process, interleaved rounds: ms per forward at M x L per knob value, outputs compared.
usage: [M=256 L=512 DTYPE=f32 ROUNDS=4 N=5] python tools/forward_knob_ab.py gemm_epi2=0,1"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import ops, synthetic
from rnamsm.model import MSATransformer
name, vals = sys.argv[1].split("=")
vals = [int(v) for v in vals.split(",")]
M, L = int(os.environ.get("M", 256)), int(os.environ.get("L", 512))
dev = "cuda:0"
model = MSATransformer(num_layers=10)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(seed=0).items()}, strict=True)
model = model.eval().to(dev)
model.gemm_dtype = os.environ.get("DTYPE", "f32")
tok = torch.from_numpy(synthetic.make_tokens(M, L, 0)).to(dev)
default = ops.get_param(name)
times, outs = {v: [] for v in vals}, {}
n = int(os.environ.get("N", 5))
try:
    for rnd in range(int(os.environ.get("ROUNDS", 4))):
        for v in vals:
            ops.set_param(name, v)
            out = model.forward_one(tok, has_padding=False)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                out = model.forward_one(tok, has_padding=False)
            b.record(); torch.cuda.synchronize()
            times[v].append(a.elapsed_time(b) / n)
            outs[v] = out["emb"].clone()
finally:
    ops.set_param(name, default)
base = vals[0]
for v in vals:
    same = bool(torch.equal(outs[v], outs[base]))
    print(f"{name}={v}: median {statistics.median(times[v]):.3f} ms  min {min(times[v]):.3f} ms  (x{statistics.median(times[base]) / statistics.median(times[v]):.4f} vs {name}={base}; emb bit-identical to it: {same})", flush=True)
