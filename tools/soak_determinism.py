#!/usr/bin/env python3
"""Soak test: the same forward N times per arithmetic mode, every output compared bit for bit with the first run.
The 16-bit kernels synchronise with hand-written s_waitcnt / s_barrier sequences around LDS-DMA; a latent race would show
up here as a rare mismatch.  ITER (default 150), shapes cfg3 and an odd mid-size one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import synthetic
from rnamsm.model import MSATransformer
ITER = int(os.environ.get("ITER", 150))
state = synthetic.make_state_dict(seed=0)
m = MSATransformer(num_layers=10)
m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
m = m.eval().cuda()
bad = 0
for (R, C), iters in (((256, 512), ITER), ((37, 300), 2 * ITER), ((64, 128), 2 * ITER)):
    toks = torch.from_numpy(synthetic.make_tokens(R, C, 5)).cuda()
    for mode in ("f32", "f16x3", "bf16"):
        m.gemm_dtype = mode
        ref = m.forward_one(toks)
        ref_emb, ref_atp = ref["emb"].clone(), ref["atp"].clone()
        mism = 0
        for i in range(iters):
            out = m.forward_one(toks)
            if not (torch.equal(out["emb"], ref_emb) and torch.equal(out["atp"], ref_atp)):
                mism += 1
        bad += mism
        print(f"{R}x{C} {mode}: {iters} reruns, {mism} mismatches", flush=True)
# round 4: the configs[4] shape in the 16-bit modes (two query blocks per wave, the operand-staged tied-row kernel) and a token-packed batch
if os.environ.get("BIG", "1") != "0":
    import numpy as np
    toks = torch.from_numpy(synthetic.make_tokens(1024, 1023, 5)).cuda()
    for mode in ("bf16", "f16x3"):
        m.gemm_dtype = mode
        ref = m.forward_one(toks, need_repr=False)
        e, a = ref["emb"].clone(), ref["atp"].clone()
        mism = 0
        for _ in range(max(4, ITER // 12)):
            o = m.forward_one(toks, need_repr=False)
            mism += 0 if (torch.equal(o["emb"], e) and torch.equal(o["atp"], a)) else 1
        bad += mism
        print(f"1024x1024 {mode}: {max(4, ITER // 12)} reruns, {mism} mismatches", flush=True)
    m.gemm_dtype = "f32"
    rng = np.random.default_rng(3)
    msas = [torch.from_numpy(synthetic.make_tokens(int(rng.integers(1, 40)), int(rng.integers(5, 150)), 100 + i)).cuda() for i in range(40)]
    ref = m.forward_packed(msas)
    mism = 0
    for _ in range(ITER):
        out = m.forward_packed(msas)
        mism += 0 if all(torch.equal(x["emb"], y["emb"]) and torch.equal(x["atp"], y["atp"]) for x, y in zip(out, ref)) else 1
    bad += mism
    print(f"packed batch of 40 unlike alignments: {ITER} reruns, {mism} mismatches", flush=True)
    # round 5: every member of that batch against its OWN forward, bit for bit (one arithmetic per alignment), and the packed batch
    # in the 16-bit modes (reruns)
    mism = sum(0 if (torch.equal(r["emb"], (o := m.forward_one(t, need_repr=False))["emb"]) and torch.equal(r["atp"], o["atp"])) else 1
               for t, r in zip(msas, ref))
    bad += mism
    print(f"packed members against their own forward: {len(msas)} alignments, {mism} not bit-identical", flush=True)
    for mode in ("bf16", "f16x3"):
        m.gemm_dtype = mode
        ref16 = m.forward_packed(msas)
        mism = 0
        for _ in range(ITER):
            out = m.forward_packed(msas)
            mism += 0 if all(torch.equal(x["emb"], y["emb"]) and torch.equal(x["atp"], y["atp"]) for x, y in zip(out, ref16)) else 1
        bad += mism
        print(f"packed batch in {mode}: {ITER} reruns, {mism} mismatches", flush=True)
    m.gemm_dtype = "f32"
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
