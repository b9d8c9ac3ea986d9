#!/usr/bin/env python3
"""16-bit modes on small and mid-size alignments: the forward with the 256x256-tile GEMM kernels from 2048 tokens on
("gemm16_big_rows_fwd" = 2048, the rule until round 3) against rnamsm_forward's token thresholds (default: 10752 plain bf16 / 8960
hi/lo) and against the 128x128 kernel everywhere ("gemm16_dma" = 1), one process.  SHAPES=MxL,..."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import _lib, synthetic
from rnamsm.model import MSATransformer
dev = torch.device("cuda:0"); lib = _lib.load()
m = MSATransformer(num_layers=10)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(seed=0).items()}, strict=True)
m = m.eval().to(dev); m.check_finite = False
def timeit(fn, n=8):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[2:])
SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SHAPES", "16x128,32x128,64x128,512x36,100x100,72x256,128x256,200x300").split(",")]
for (M, L) in SHAPES:
    toks = torch.from_numpy(synthetic.make_tokens(M, L, 0)).to(dev)
    line = f"M={M} L={L} ({M * L} tokens):"
    for mode in ("bf16", "f16x3"):
        m.gemm_dtype = mode
        res = {}
        out = {}
        for rnd in range(2):
            for tag, knobs in (("default", {}), ("256 from 2048", {"gemm16_big_rows_fwd": 2048}), ("128 only", {"gemm16_dma": 1})):
                _lib.check(lib.rnamsm_set_param(b"gemm16_dma", 3)); _lib.check(lib.rnamsm_set_param(b"gemm16_big_rows_fwd", 0))
                for k, v in knobs.items():
                    _lib.check(lib.rnamsm_set_param(k.encode(), v))
                o = m.forward_one(toks); torch.cuda.synchronize()
                out[tag] = o["emb"].clone()
                res.setdefault(tag, []).append(timeit(lambda: m.forward_one(toks)))
        a, b, c = min(res["default"]), min(res["256 from 2048"]), min(res["128 only"])
        same = bool(torch.equal(out["default"], out["256 from 2048"]))
        line += f" | {mode}: default {a:.2f} ms, 256x256 from 2048 rows {b:.2f} (x{b / a:.3f}), 128x128 only {c:.2f}; emb identical {same}"
    _lib.check(lib.rnamsm_set_param(b"gemm16_big_rows_fwd", 0))
    _lib.check(lib.rnamsm_set_param(b"gemm16_dma", 3))
    print(line, flush=True)
