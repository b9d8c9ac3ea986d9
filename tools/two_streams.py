#!/usr/bin/env python3
"""Do two independent MSAs in flight on two HIP streams (own workspaces) beat one after the other?  The kernels of one MSA
form a dependency chain; a second queue can fill the first one's kernel tails (last, partially filled round of blocks) and
run its HBM-bound kernels (LayerNorm, softmax) beside the other's MFMA-bound ones."""
import os, sys, time, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import synthetic
from rnamsm.model import MSATransformer
dev = "cuda:0"
M, L = int(os.environ.get("M", 256)), int(os.environ.get("L", 512))
N = int(os.environ.get("N", 8))
mode = os.environ.get("MODE", "f32")
m1 = MSATransformer(num_layers=10)
m1.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(seed=0).items()}, strict=True)
m1 = m1.eval().to(dev)
m2 = copy.deepcopy(m1)
for m in (m1, m2):
    m.gemm_dtype = mode
    m.check_finite = False
toks = [torch.from_numpy(synthetic.make_tokens(M, L, i)).to(dev) for i in range(N)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for m in (m1, m2):
    m.forward_one(toks[0], has_padding=False)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    outs = [m1.forward_one(t, has_padding=False) for t in toks]
    torch.cuda.synchronize()
    seq = time.perf_counter() - t0
    t0 = time.perf_counter()
    outs2 = []
    for i, t in enumerate(toks):
        m, s = ((m1, s1), (m2, s2))[i & 1]
        with torch.cuda.stream(s):
            outs2.append(m.forward_one(t, has_padding=False))
    torch.cuda.synchronize()
    par = time.perf_counter() - t0
    same = all(torch.equal(a["emb"], b["emb"]) and torch.equal(a["atp"], b["atp"]) for a, b in zip(outs, outs2))
    print(f"{mode} M={M} L={L}: sequential {N * M * L / seq:9.0f} res/s ({1e3 * seq / N:.2f} ms/MSA)   two streams {N * M * L / par:9.0f} res/s "
          f"({1e3 * par / N:.2f} ms/MSA)   x{seq / par:.3f}   identical {same}", flush=True)
