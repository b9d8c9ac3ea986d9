#!/usr/bin/env python3
"""End-to-end error and speed of the GEMM arithmetic modes (f32 / f16x3 / bf16) of rnamsm_forward:
error vs the reference fixtures (fp32 and fp64 runs of the reference) and vs the CPU oracle at M=64, L=128."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from rnamsm import synthetic
from rnamsm.model import MSATransformer
from oracle import msm_oracle as O
state = synthetic.make_state_dict(seed=0)
m = MSATransformer(num_layers=10); m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}); m = m.eval().cuda()
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
G = os.path.join(ROOT, "tests", "golden")
toks64 = synthetic.make_tokens(64, 128, 0)
res = O.forward(torch.from_numpy(toks64), O.to_torch_params(state, torch.float64))
o_emb, o_atp = (t.numpy() for t in O.pack_outputs(res))
from rnamsm import _lib
for mode, attn16 in (("f32", 1), ("f16x3", 0), ("f16x3", 1), ("bf16", 0), ("bf16", 1)):
    m.gemm_dtype = mode
    _lib.check(_lib.load().rnamsm_set_param(b"attn16", attn16))
    line = [f"{mode} attn16={attn16}"]
    for name in ("m8_c17", "m16_c33"):
        g, g64 = np.load(f"{G}/forward_{name}.npz"), np.load(f"{G}/forward_{name}_fp64.npz")
        out = m.forward_one(torch.from_numpy(g["tokens"]).cuda())
        e, a = out["emb"].cpu().numpy(), out["atp"].cpu().numpy()
        line.append(f"{name}: emb rel vs ref32 {rel(e, g['emb']):.1e} vs ref64 {rel(e, g64['emb']):.1e} maxabs/max {np.abs(e-g64['emb']).max()/np.abs(g64['emb']).max():.1e} atp maxabs {np.abs(a - g64['atp']).max():.1e}")
    out = m.forward_one(torch.from_numpy(toks64).cuda())
    e, a = out["emb"].cpu().numpy(), out["atp"].cpu().numpy()
    line.append(f"M64xL128 vs oracle fp64: emb rel {rel(e, o_emb):.1e} atp maxabs {np.abs(a - o_atp).max():.1e}")
    t = torch.from_numpy(synthetic.make_tokens(256, 512, 0)).cuda()
    for _ in range(2): m.forward_one(t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): m.forward_one(t)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
    line.append(f"cfg3 {dt*1e3:.1f} ms {256*512/dt:.0f} res/s")
    print("\n   ".join(line))
