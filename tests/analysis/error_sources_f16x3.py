#!/usr/bin/env python3
"""Which part of the f16x3 mode sets its distance from the fp64 truth?  Layer-wise path with f16x3 switched on for one
module class at a time (the others exact fp32)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "rna-msm_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import truth
from rnamsm import modules as M_, synthetic
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
    res = m(t[None], repr_layers=[0, 10], need_head_weights=True)
    emb = res["representations"][10][0, 0, 1:]
    atp = res["row_attentions"][0][..., 1:, 1:].reshape(-1, L - 1, L - 1)
    e = truth.errors(emb, atp, t_emb, t_atp)
    print(json.dumps({label: {k: float(f"{v:.3e}") for k, v in e.items()}}), flush=True)


run("all f32")
for name, cls in (("row attention", M_.RowSelfAttention), ("column attention", M_.ColumnSelfAttention), ("ffn", M_.FeedForwardNetwork)):
    for mod in m.modules():
        if hasattr(mod, "gemm_dtype") and mod is not m:
            mod.gemm_dtype = "f16x3" if isinstance(mod, cls) else "f32"
    run(f"f16x3 in {name} only")
m.gemm_dtype = "f16x3"
run("all f16x3")
