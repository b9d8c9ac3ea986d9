#!/usr/bin/env python3
"""Where a LONE small forward spends its time: wall per forward (hooks off) against the per-kernel sums of one instrumented forward,
per arithmetic mode.  SHAPES=8x64,16x128 (rows x columns-without-<cls>)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import synthetic, _lib
from rnamsm.model import MSATransformer
dev = "cuda:0"
model = MSATransformer(num_layers=10)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(seed=0).items()}, strict=True)
model = model.eval().to(dev)
lib = _lib.load()
for kv in os.environ.get("KNOBS", "").split(","):          # KNOBS=gemm_splitk_short=0
    if kv:
        _lib.check(lib.rnamsm_set_param(kv.split("=")[0].encode(), int(kv.split("=")[1])))
for shp in os.environ.get("SHAPES", "8x64,16x128,32x128,64x128").split(","):
    r, l = (int(v) for v in shp.split("x"))
    t = torch.from_numpy(synthetic.make_tokens(r, l, 1)).to(dev)
    for mode in os.environ.get("MODES", "f32,bf16").split(","):
        model.gemm_dtype = mode
        for _ in range(3):
            model.forward_one(t, has_padding=False, need_repr=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            model.forward_one(t, has_padding=False, need_repr=False)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e3
        lib.rnamsm_timing_reset(); lib.rnamsm_timing_enable(1)
        model.forward_one(t, has_padding=False, need_repr=False); torch.cuda.synchronize()
        kt = _lib.kernel_timings(); lib.rnamsm_timing_enable(0)
        tot = sum(v["ms"] for v in kt.values()); n = sum(v["launches"] for v in kt.values())
        print(f"{r} x {l + 1} {mode}: wall {wall:.2f} ms; kernels {tot:.2f} ms in {n} launches | " +
              " ".join(f"{k} {v['ms']:.2f}/{v['launches']}" for k, v in kt.items() if v["launches"]), flush=True)
    model.gemm_dtype = "f32"
