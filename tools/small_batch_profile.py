#!/usr/bin/env python3
"""Where a batch of small MSAs spends its time: per-kernel HIP-event timings (rnamsm_timing_*) of rnamsm_forward_batch for
B same-shape alignments, next to one lone forward of the same shape.  SHAPES="B:R:C,..." (default 32:8:64,16:16:128,8:32:128)."""
import os, sys, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import numpy as np, torch
from rnamsm import _lib, synthetic
from rnamsm.model import MSATransformer
dev = torch.device("cuda:0")
lib = _lib.load()
state = synthetic.make_state_dict(seed=0)
m = MSATransformer(num_layers=10)
m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
m = m.eval().to(dev)
for spec in os.environ.get("SHAPES", "32:8:64,16:16:128,8:32:128").split(","):
    B, R, C = (int(x) for x in spec.split(":"))
    st = torch.from_numpy(np.stack([synthetic.make_tokens(R, C, 900 + b) for b in range(B)])).to(dev)
    for _ in range(3):
        m.forward_batch(st, has_padding=False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); m.forward_batch(st, has_padding=False); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    wall = statistics.median(ts)
    lib.rnamsm_timing_reset(); lib.rnamsm_timing_enable(1)
    m.forward_batch(st, has_padding=False); torch.cuda.synchronize()
    lib.rnamsm_timing_enable(0)
    tim = _lib.kernel_timings()
    tot = sum(v["ms"] for v in tim.values())
    print(f"B={B} R={R} C={C}: wall {1e3 * wall:.3f} ms = {B * R * C / wall / 1e3:.0f} k residues/s; kernels {tot:.3f} ms ({tot / (1e3 * wall):.2f} of wall)")
    for k, v in tim.items():
        if v["launches"]:
            print(f"    {k:13s} {v['launches']:4d} launches {v['ms']:8.3f} ms  avg {1e3 * v['ms'] / v['launches']:7.1f} us  bound {v['bound_ms']:.3f} ms")
