#!/usr/bin/env python3
"""f16x3: distance from the fp64 truth as a function of the row_logits16 slab cap (knob row16_max_rows)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "rna-msm_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import truth
from rnamsm import _lib, synthetic
from rnamsm.model import MSATransformer
M, L = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (256, 512)))
dev = "cuda:0"
lib = _lib.load()
toks = synthetic.make_tokens(M, L, 0)
m = MSATransformer(num_layers=10)
m.load_state_dict({k: torch.from_numpy(v) for k, v in truth.state().items()}, strict=True)
m = m.eval().to(dev)
t_emb, t_atp = truth.oracle_outputs(toks, torch.float64, dev)
t = torch.from_numpy(toks).to(dev)
for mode in ("f16x3",):
    m.gemm_dtype = mode
    for cap in (0, 128, 64, 32, 16, 8):
        _lib.check(lib.rnamsm_set_param(b"row16_max_rows", cap))
        m._workspace = None
        torch.cuda.synchronize()
        import time
        out = m.forward_one(t); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = m.forward_one(t); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        e = truth.errors(out["emb"], out["atp"], t_emb, t_atp)
        print(json.dumps({f"{mode} cap {cap}": {k: float(f"{v:.3e}") for k, v in e.items()}, "ms": round(1e3 * dt, 2)}), flush=True)
_lib.check(lib.rnamsm_set_param(b"row16_max_rows", 0))
