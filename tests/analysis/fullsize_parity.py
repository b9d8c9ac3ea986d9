#!/usr/bin/env python3
"""Full-size parity run: the M=256 x L=512 forward (BASELINE configs[2]) on the GPU against the CPU oracle on the box's
host cores (all ten layers, ~6 min of CPU time, ~20 GB of RAM), for the exact path and the f16x3 mode, next to the exact
path's own re-ordering noise.  Prints one JSON line.  The oracle is fp32 on the CPU, i.e. a different summation order of
the same arithmetic: at this depth the synthetic problem amplifies that alone to ~1e-4 / ~1e-3."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from rnamsm import synthetic
from rnamsm.model import MSATransformer
from oracle import msm_oracle as O
M, L = int(os.environ.get("M", 256)), int(os.environ.get("L", 512))
torch.set_num_threads(os.cpu_count())
state = synthetic.make_state_dict(seed=0)
toks = synthetic.make_tokens(M, L, 0)
t0 = time.perf_counter()
emb, atp = O.pack_outputs(O.forward(torch.from_numpy(toks), O.to_torch_params(state)))
t_cpu = time.perf_counter() - t0
emb, atp = emb.numpy(), atp.numpy()
m = MSATransformer(num_layers=10)
m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
m = m.eval().cuda()
g = torch.from_numpy(toks).cuda()
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
res = {"shape": [M, L], "oracle_seconds": t_cpu, "oracle": "oracle/msm_oracle.py, torch CPU fp32, %d threads" % os.cpu_count()}
for mode in ("f32", "f16x3"):
    m.gemm_dtype = mode
    out = m.forward_one(g)
    e, a = out["emb"].cpu().numpy(), out["atp"].cpu().numpy()
    res[mode] = {"emb_rel_l2": rel(e, emb), "emb_max_abs_over_max": float(np.abs(e - emb).max() / np.abs(emb).max()),
                 "atp_max_abs": float(np.abs(a - atp).max()), "atp_mean_abs": float(np.abs(a - atp).mean())}
m.gemm_dtype = "f32"
perm = torch.cat([torch.zeros(1, dtype=torch.long), 1 + torch.randperm(M - 1, generator=torch.Generator().manual_seed(0))]).cuda()
ref = m.forward_one(g); per = m.forward_one(g[perm])
res["f32_vs_itself_rows_permuted"] = {"emb_rel_l2": rel(per["emb"].cpu().numpy(), ref["emb"].cpu().numpy()),
                                       "atp_max_abs": float((per["atp"] - ref["atp"]).abs().max()),
                                       "atp_mean_abs": float((per["atp"] - ref["atp"]).abs().mean())}
print(json.dumps(res))
