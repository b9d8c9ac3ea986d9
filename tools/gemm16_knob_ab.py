#!/usr/bin/env python3
"""Plain-bf16 plane GEMMs under knob settings, interleaved rounds in one process (the pool's devices differ and the chip
is power-managed: only same-process numbers compare).  CONFIGS="base;gemm16_mfma16=2;gemm16_mfma16=0,gemm_group=1"
(first = the reference for the diff and the ratio), T=..., ROUNDS=..., FORWARD=1 adds the whole bf16 forward at M=256 L=512."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import _lib, ops, synthetic
from rnamsm._lib import ACT_GELU_ERF, ACT_NONE
ROUNDS = int(os.environ.get("ROUNDS", 3))
OPERAND_SCALE = float(os.environ.get("OPERAND_SCALE", 1.0))      # 0 = zero-filled A and W (the clock the chip holds depends on the data: power)
MODE = os.environ.get("MODE", "bf16")                     # bf16 (one plane per operand) | f16x3 (fp16 hi/lo planes, 3 products)
LO, FMT = MODE == "f16x3", 1 if MODE == "f16x3" else 0
dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(0)


def parse(c):
    return {} if c.strip() in ("", "base") else {k: int(v) for k, v in (kv.split("=") for kv in c.split(","))}


CONFIGS = [(c, parse(c)) for c in os.environ.get("CONFIGS", "base;gemm16_mfma16=2").split(";")]
KNOBS = sorted({k for _, d in CONFIGS for k in d})
DEFAULTS = {k: lib.rnamsm_get_param(k.encode()) for k in KNOBS}


def apply(d):
    for k in KNOBS:
        _lib.check(lib.rnamsm_set_param(k.encode(), d.get(k, DEFAULTS[k])))


def timeit(fn, n=8):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[2:])


for T in [int(x) for x in os.environ.get("T", "131072").split(",")]:
    tot = [0.0] * len(CONFIGS)
    for tag, N, K, act, res, opl, per_layer in [("qkv", 2304, 768, ACT_NONE, False, True, 2), ("out", 768, 768, ACT_NONE, True, False, 2),
                                                ("fc1", 3072, 768, ACT_GELU_ERF, False, True, 1), ("fc2", 768, 3072, ACT_NONE, True, False, 1)]:
        a = torch.randn(T, K, device=dev) * (0.5 if LO else 1.0) * OPERAND_SCALE; w = torch.randn(N, K, device=dev) * 0.04 * OPERAND_SCALE; b = torch.randn(N, device=dev) * 0.05
        r = torch.randn(T, N, device=dev) if res else None
        fl = 2.0 * T * N * K
        ap = ops.split_bf16(a, want_lo=LO, fmt=FMT); wp = ops.split_bf16(w, want_lo=LO, fmt=FMT)
        del a
        times, outs = [[] for _ in CONFIGS], [None] * len(CONFIGS)
        out = None if opl else torch.empty(T, N, device=dev)
        fn = lambda: ops.linear_planes(ap, wp, b, act=act, residual=r, out=out, out_planes=opl, fmt=FMT)
        for rnd in range(ROUNDS):
            for i, (_, d) in enumerate(CONFIGS):
                apply(d)
                res_t = fn(); torch.cuda.synchronize()
                if opl:                                   # the value the planes hold (hi, or hi + lo)
                    ht = torch.float16 if FMT == 1 else torch.bfloat16
                    outs[i] = res_t[0].view(ht).float() + (res_t[1].view(ht).float() if res_t[1] is not None else 0.0)
                else:
                    outs[i] = res_t.clone()
                times[i].append(timeit(fn))
        med = [statistics.median(t) for t in times]
        for i in range(len(CONFIGS)):
            tot[i] += per_layer * med[i]
        line = f"{MODE} T={T} {tag:4s}"
        for i, (name, _) in enumerate(CONFIGS):
            diff = float((outs[0].float() - outs[i].float()).norm() / outs[0].float().norm())       # relative L2
            line += f" | {name}: {med[i]:.3f} ms {fl / med[i] / 1e9:5.0f} TF x{med[0] / med[i]:.3f} diff {diff:.2g}"
        print(line, flush=True)
        del ap, wp, r, out, outs
        torch.cuda.empty_cache()
    print(f"T={T} six GEMMs of a layer: " + " | ".join(f"{n}: {t:.3f} ms x{tot[0] / t:.3f}" for (n, _), t in zip(CONFIGS, tot)), flush=True)

if os.environ.get("FORWARD", "1") != "0":
    from rnamsm.model import MSATransformer
    state = synthetic.make_state_dict(seed=0)
    m = MSATransformer(num_layers=10)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    m = m.eval().to(dev); m.gemm_dtype = MODE; m.check_finite = False
    toks = torch.from_numpy(synthetic.make_tokens(256, 512, 0)).to(dev)
    res, embs = [[] for _ in CONFIGS], [None] * len(CONFIGS)
    for rnd in range(3):
        for i, (_, d) in enumerate(CONFIGS):
            apply(d)
            o = m.forward_one(toks); torch.cuda.synchronize()
            embs[i] = o["emb"].clone()
            res[i].append(timeit(lambda: m.forward_one(toks), n=5))
    med = [statistics.median(t) for t in res]
    print(f"forward {MODE} M=256 L=512: " + " | ".join(
        f"{n}: {t:.2f} ms x{med[0] / t:.3f} emb diff {float((embs[0] - embs[i]).abs().max()):.2g}" for i, ((n, _), t) in enumerate(zip(CONFIGS, med))), flush=True)
apply({})
