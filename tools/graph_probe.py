#!/usr/bin/env python3
"""Does a captured HIP graph of the whole forward beat the plain stream of launches for small MSAs?  At M=64 L=128 the
bench reports 5 % of a step outside kernels (about 140 launches of ~90 us each).  The probe captures one forward (the same
rnamsm_forward call, static token / output buffers) and replays it; outputs must be bit-identical."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import synthetic
from rnamsm.model import MSATransformer
dev = "cuda:0"
mode = os.environ.get("MODE", "f32")
N = int(os.environ.get("N", 50))
model = MSATransformer(num_layers=10)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(seed=0).items()}, strict=True)
model = model.eval().to(dev)
model.gemm_dtype = mode
model.check_finite = False
for M, L in [(64, 128), (32, 64), (512, 36), (128, 256), (256, 512)]:
    tok = torch.from_numpy(synthetic.make_tokens(M, L, 0)).to(dev)
    ref = model.forward_one(tok, has_padding=False)            # also builds every cache outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        model.forward_one(tok, has_padding=False)
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            out = model.forward_one(tok, has_padding=False)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(ref["emb"], out["emb"]) and torch.equal(ref["atp"], out["atp"])
    n = max(3, N * 8192 // (M * L))
    best = {}
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            model.forward_one(tok, has_padding=False)
        torch.cuda.synchronize()
        plain = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / n
        best["plain"] = min(best.get("plain", 1e9), plain)
        best["graph"] = min(best.get("graph", 1e9), graph)
    print(f"{mode} M={M} L={L}: launches {1e3 * best['plain']:.3f} ms/MSA   graph replay {1e3 * best['graph']:.3f} ms/MSA   "
          f"x{best['plain'] / best['graph']:.3f}   identical {same}", flush=True)
    del g
