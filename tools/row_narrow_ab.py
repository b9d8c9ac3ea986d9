#!/usr/bin/env python3
"""K4 / K5 / K6 at C <= 64: the narrow kernels (knob row_narrow = 1) against the 128 x 128 tile kernels (0), one process, interleaved.
usage: [SHAPES=512x36,64x40,300x17,128x64 ROUNDS=5 N=20 RPB=0,4,8,16] python tools/row_narrow_ab.py"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import ops
dev = "cuda:0"
H, D = 12, 768
N, ROUNDS = int(os.environ.get("N", 20)), int(os.environ.get("ROUNDS", 5))
RPB = [int(x) for x in os.environ.get("RPB", "0").split(",")]


def timed(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(N):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / N * 1000.0


for shape in os.environ.get("SHAPES", "512x36,64x40,300x17,128x64,1024x60").split(","):
    R, C = (int(x) for x in shape.split("x"))
    torch.manual_seed(0)
    g = torch.randn(R * C, 3 * D, device=dev)
    res = {}
    try:
        for rnd in range(ROUNDS):
            for narrow in (0, 1):
                ops.set_param("row_narrow", narrow)
                partial, ns = ops.row_logits(g[:, :D], g[:, D:2 * D], R, C, H)
                probs = ops.softmax_rows(partial, logit_scale=ops.depth_scaling(R))
                res.setdefault(("logits", narrow), []).append(timed(lambda: ops.row_logits(g[:, :D], g[:, D:2 * D], R, C, H)))
                res.setdefault(("softmax", narrow), []).append(timed(lambda: ops.softmax_rows(partial, logit_scale=ops.depth_scaling(R))))
                res.setdefault(("apply", narrow, 0), []).append(timed(lambda: ops.row_apply(probs, g[:, 2 * D:], R, C, H)))
    finally:
        ops.set_param("row_narrow", 1)
    hbm = 2.0 * R * C * H * 64 * 4 / 1e6
    print(f"{R} x {C} (nsplit {ns}; q+k or v+ctx = {hbm:.1f} MB = {hbm / 6.3:.1f} us at 6.3 TB/s): " +
          " | ".join(f"{k[0]}{'' if len(k) < 3 or not k[2] else '/rpb' + str(k[2])} {'narrow' if k[1] else 'tile'} {statistics.median(v):.1f} us"
                     for k, v in sorted(res.items())), flush=True)
