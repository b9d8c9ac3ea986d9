#!/usr/bin/env python3
"""BASELINE configs[4]'s "tied-row subsampling path": an alignment deeper than the model's 1024 rows is cut to 1024 rows by
the reference's greedy max-mean-Hamming rule (utils/align.py:128-148) on the device before the forward.  Times
rnamsm_greedy_select on synthetic alignments (random bases with per-row mutation of a seed sequence) and the host (numpy)
version of the same rule on a bounded sample; checks that both pick the same rows.

    python tools/greedy_select_timing.py [L [N ...]]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))

import numpy as np
import torch

from rnamsm import msa, ops

L = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
depths = [int(v) for v in sys.argv[2:]] or [2048, 8192, 32768]
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
SEL = int(os.environ.get("SELECT", 1024))
res = {"L": L, "select": SEL, "runs": []}
for N in depths:
    seed_seq = rng.randint(4, 8, size=L)
    toks = np.where(rng.rand(N, L) < 0.3, rng.randint(4, 11, size=(N, L)), seed_seq[None]).astype(np.uint8)
    u8 = torch.from_numpy(toks).to(dev)
    times = {}
    picks = {}
    for fused in (0, 2):
        ops.set_param("greedy_fused", fused)
        ops.greedy_select(u8, SEL, "max")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        picks[fused] = ops.greedy_select(u8, SEL, "max")
        torch.cuda.synchronize()
        times[fused] = time.perf_counter() - t0
    ops.set_param("greedy_fused", 1)
    idx, dt = picks[2], times[2]
    run = {"N": N, "one_launch_per_step_ms": round(1e3 * dt, 2), "three_launches_per_step_ms": round(1e3 * times[0], 2),
           "same_rows_either_way": bool(torch.equal(picks[0], picks[2])),
           "compare_bytes": (SEL - 1) * N * L, "compare_GBps": round((SEL - 1) * N * L / min(times.values()) / 1e9, 1),
           "history_bytes": int((SEL - 1) * SEL / 2 * N * 2)}
    if N <= 2048:                                   # the host rule is O(steps^2 N) in numpy: bounded sample only
        full = np.concatenate([np.zeros((N, 1), np.int64), toks.astype(np.int64)], 1)
        t0 = time.perf_counter()
        want = msa.greedy_select(full, SEL, "max")
        run["host_numpy_s"] = round(time.perf_counter() - t0, 2)
        run["same_rows_as_host"] = bool(np.array_equal(np.sort(idx.cpu().numpy()), want))
    res["runs"].append(run)
    print(json.dumps(run), flush=True)
print(json.dumps(res))
