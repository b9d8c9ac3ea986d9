#!/usr/bin/env python3
"""In-process A/B of the plane-input 16-bit GEMM staging (gemm16_dma = 0 register-staged, 1 LDS-DMA) at cfg3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import synthetic, _lib
from rnamsm.model import MSATransformer
lib = _lib.load()
state = synthetic.make_state_dict(seed=0)
m = MSATransformer(num_layers=10); m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}); m = m.eval().cuda()
t = torch.from_numpy(synthetic.make_tokens(256, 512, 0)).cuda()
for mode in ("f16x3", "bf16"):
    m.gemm_dtype = mode
    outs = {}
    for rep in range(2):
        for dma in (1, 2):
            lib.rnamsm_set_param(b"gemm16_dma", dma)
            for _ in range(2): o = m.forward_one(t)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4): o = m.forward_one(t)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
            outs[dma] = o["emb"].clone()
            print(f"{mode} dma={dma}: {dt*1e3:.1f} ms  {256*512/dt:.0f} res/s")
    print(mode, "emb max |dma2 - dma1|:", float((outs[2] - outs[1]).abs().max()))
