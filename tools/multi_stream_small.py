#!/usr/bin/env python3
"""Small MSAs leave most of the chip idle (a 256-token forward has 36-block GEMMs on 256 CUs and still costs 5.5 ms: every
one of its ~140 dependent launches lasts a block's serial time).  Do S independent MSAs in flight on S HIP streams (own
workspaces) recover it?   M=8 L=64 N=32 python tools/multi_stream_small.py"""
import os, sys, time, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import synthetic
from rnamsm.model import MSATransformer
dev = "cuda:0"
N = int(os.environ.get("N", 32))
mode = os.environ.get("MODE", "f32")
base = MSATransformer(num_layers=10)
base.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(seed=0).items()}, strict=True)
base = base.eval().to(dev)
base.gemm_dtype = mode
base.check_finite = False
models = [base] + [copy.deepcopy(base) for _ in range(7)]
streams = [torch.cuda.Stream() for _ in range(8)]
for M, L in ((4, 64), (8, 64), (16, 128), (32, 128), (64, 128), (128, 256)):
    toks = [torch.from_numpy(synthetic.make_tokens(M, L, i)).to(dev) for i in range(N)]
    for m in models:
        m.forward_one(toks[0], has_padding=False)
    torch.cuda.synchronize()
    line = []
    ref = None
    for S in (1, 2, 4, 8):
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            outs = []
            for i, t in enumerate(toks):
                with torch.cuda.stream(streams[i % S]):
                    outs.append(models[i % S].forward_one(t, has_padding=False))
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        if ref is None:
            ref = [o["emb"].clone() for o in outs]
        same = all(torch.equal(a, o["emb"]) for a, o in zip(ref, outs))
        line.append(f"{S} stream(s): {1e3 * best / N:.3f} ms/MSA ({N * M * L / best:8.0f} res/s){'' if same else ' DIFFERENT'}")
    print(f"{mode} M={M} L={L}: " + "   ".join(line), flush=True)
