#!/usr/bin/env python3
"""Token-packed batches (rnamsm_forward_packed) against the framed ragged batch and the one-by-one loop, as a function of the
batch's token count: n alignments of rlo..rhi rows x clo..chi columns drawn with a fixed seed, n grown until the batch holds
~TOKENS tokens.  Per setting: ms per batch, real residues/s, and -- KERNELS=1 -- the per-kernel split of the packed batch.
    python tools/packed_batch_timing.py            TOKENS=4096,8192,16384,32768,65536  POP=small|tiny|mid"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import numpy as np
import torch
from rnamsm import synthetic, _lib
from rnamsm.model import MSATransformer
dev = "cuda:0"
model = MSATransformer(num_layers=10)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(seed=0).items()}, strict=True)
model = model.eval().to(dev)
POPS = {"tiny": ((2, 12), (41, 81)), "small": ((4, 24), (41, 121)), "mid": ((16, 64), (60, 200))}
(rlo, rhi), (clo, chi) = POPS[os.environ.get("POP", "small")]
lib = _lib.load()
for kv in os.environ.get("KNOBS", "").split(","):          # KNOBS=ln_fold=3,gemm_tile=2: rnamsm_set_param before the runs
    if kv:
        _lib.check(lib.rnamsm_set_param(kv.split("=")[0].encode(), int(kv.split("=")[1])))
for target in [int(v) for v in os.environ.get("TOKENS", "4096,8192,12288,16384,20480,24576,32768,49152,65536").split(",")]:
    rng = np.random.default_rng(0)
    shapes, real = [], 0
    while real < target:
        r, c = int(rng.integers(rlo, rhi + 1)), int(rng.integers(clo, chi + 1))
        shapes.append((r, c)); real += r * c
    msas = [torch.from_numpy(synthetic.make_tokens(r, c - 1, i)).to(dev) for i, (r, c) in enumerate(shapes)]
    n = len(msas)
    frame = n * max(r for r, _ in shapes) * max(c for _, c in shapes)
    def best(fn, reps=4):
        fn(); fn(); torch.cuda.synchronize()
        b = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
        return b
    t_pk = best(lambda: model.forward_packed(msas))
    t_fr = best(lambda: model.forward_ragged(msas, packed=False)) if frame <= 4 * real and frame <= 262144 else float("nan")
    t_one = best(lambda: [model.forward_one(t, has_padding=False, need_repr=False) for t in msas], reps=2) if n <= 128 else float("nan")
    line = (f"{n:4d} MSAs {rlo}-{rhi} x {clo}-{chi}, {real:6d} tokens (frame {frame:7d}): packed {1e3 * t_pk:7.2f} ms = {real / t_pk:8.0f} res/s; "
            f"framed {1e3 * t_fr:7.2f} ms; one by one {1e3 * t_one:7.2f} ms")
    if os.environ.get("KERNELS"):
        lib.rnamsm_timing_reset(); lib.rnamsm_timing_enable(1)
        model.forward_packed(msas); torch.cuda.synchronize()
        kt = _lib.kernel_timings(); lib.rnamsm_timing_enable(0)
        line += " | " + " ".join(f"{k} {v['ms']:.2f}" for k, v in kt.items() if v["launches"])
    print(line, flush=True)
