#!/usr/bin/env python3
"""A/B of the one-wave-per-(column, head) column kernels against the 128-query-block kernels (knob "col_small": 1 / 0) in one
process, interleaved rounds, on the prescaled entry the exact forward calls.  Shipped: R <= 16 (col_attn_small_kernel).  Round 6 ran
this script with a second kernel for R = 17..64 (col_attn_wave_kernel, `git show 1a3ce62:rna-msm_amd/csrc/col_attn.hip`): correct, not
faster, removed (profiles/r06_col_wave_ab.log, EXPERIMENTS R6.6).  SHAPES=64x128,... ; FORWARD=1 adds the exact forward at those shapes."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import _lib, ops, synthetic
H = 12
D = 64 * H
dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(0)
SHAPES = [tuple(int(x) for x in s.split("x")) for s in os.environ.get("SHAPES", "64x128,32x128,17x64,48x300,64x512,33x40,24x36,8x64").split(",")]


def timeit(fn, n=9):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[2:])


for R, C in SHAPES:
    qkv = torch.randn(R * C, 3 * D, device=dev)
    qkv[:, :D] *= 0.125 * 1.5
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    res, outs = {0: [], 1: []}, {}
    for rnd in range(3):
        for knob in (0, 1):
            _lib.check(lib.rnamsm_set_param(b"col_small", knob))
            ctx = torch.empty(R * C, D, device=dev)
            res[knob].append(timeit(lambda: ops.col_attn(q, k, v, R, C, H, out=ctx, prescaled=True)))
            outs[knob] = ctx
    a, b = statistics.median(res[0]), statistics.median(res[1])
    diff = float((outs[0] - outs[1]).norm() / outs[0].norm())
    hbm = 4.0 * 4 * R * C * D / 1e6
    print(f"R={R:3d} C={C:4d}  blocks {1e3 * a:7.1f} us   one wave per problem {1e3 * b:7.1f} us   x{a / b:.2f}   rel diff {diff:.1e}   "
          f"(q,k,v,ctx = {hbm:.1f} MB = {hbm / 6.3:.1f} us at 6.3 TB/s)", flush=True)
_lib.check(lib.rnamsm_set_param(b"col_small", 1))
if os.environ.get("FORWARD", "1") != "0":
    from rnamsm.model import MSATransformer
    state = synthetic.make_state_dict(seed=0)
    m = MSATransformer(num_layers=10)
    m.load_state_dict({k_: torch.from_numpy(v_) for k_, v_ in state.items()}, strict=True)
    m = m.eval().to(dev); m.check_finite = False
    for R, C in SHAPES:
        toks = torch.from_numpy(synthetic.make_tokens(R, C, 0)).to(dev)
        res = {0: [], 1: []}
        for rnd in range(3):
            for knob in (0, 1):
                _lib.check(lib.rnamsm_set_param(b"col_small", knob))
                m.forward_one(toks); torch.cuda.synchronize()
                res[knob].append(timeit(lambda: m.forward_one(toks), n=7))
        a, b = statistics.median(res[0]), statistics.median(res[1])
        print(f"forward {R:3d} x {C:4d}: blocks {a:.3f} ms   one wave per problem {b:.3f} ms   x{a / b:.3f}   ({R * C / b:.0f} k residues/s)", flush=True)
    _lib.check(lib.rnamsm_set_param(b"col_small", 1))
