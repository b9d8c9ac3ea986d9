#!/usr/bin/env python3
"""Plain-bf16 plane GEMMs: the 256x256 kernels (gemm16_q16 / gemm16_swp, knob gemm16_pp=0) vs the epilogue-hiding kernel
(gemm16_pp_kernel, gemm16_pp=1) at the M=256 L=512 shapes (T = 131072 tokens) and at T = 1048576 (M=L=1024), interleaved
rounds in one process; then the whole bf16 forward with the knob off / on.  T=... ROUNDS=... env."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch
from rnamsm import _lib, ops, synthetic
from rnamsm._lib import ACT_GELU_ERF, ACT_NONE
ROUNDS = int(os.environ.get("ROUNDS", 3))
dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(0)


def timeit(fn, n=8):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[2:])


def setpp(v):
    _lib.check(lib.rnamsm_set_param(b"gemm16_pp", v))


for T in [int(x) for x in os.environ.get("T", "131072,1048576").split(",")]:
    tot = {0: 0.0, 1: 0.0}
    for tag, N, K, act, res, opl, per_layer in [("qkv", 2304, 768, ACT_NONE, False, True, 2), ("out", 768, 768, ACT_NONE, True, False, 2),
                                                ("fc1", 3072, 768, ACT_GELU_ERF, False, True, 1), ("fc2", 768, 3072, ACT_NONE, True, False, 1)]:
        a = torch.randn(T, K, device=dev); w = torch.randn(N, K, device=dev) * 0.04; b = torch.randn(N, device=dev) * 0.05
        r = torch.randn(T, N, device=dev) if res else None
        fl = 2.0 * T * N * K
        ap = ops.split_bf16(a, want_lo=False); wp = ops.split_bf16(w, want_lo=False)
        del a
        times, outs = {0: [], 1: []}, {}
        out = None if opl else torch.empty(T, N, device=dev)
        fn = lambda: ops.linear_planes(ap, wp, b, act=act, residual=r, out=out, out_planes=opl)
        for rnd in range(ROUNDS):
            for v in (0, 1):
                setpp(v)
                res_t = fn(); torch.cuda.synchronize()
                outs[v] = (res_t[0] if opl else res_t).clone()
                times[v].append(timeit(fn))
        t0, t1 = statistics.median(times[0]), statistics.median(times[1])
        tot[0] += per_layer * t0; tot[1] += per_layer * t1
        hbm = (2.0 * (T * K + N * K) + (2.0 if opl else 4.0) * T * N + (4.0 * T * N if res else 0.0)) / 6.3e12 * 1e3
        print(f"T={T} {tag:4s} old {t0:.3f} ms {fl / t0 / 1e9:5.0f} TF | pp {t1:.3f} ms {fl / t1 / 1e9:5.0f} TF x{t0 / t1:.3f} | bound mfma {fl / 2.5e15 * 1e3:.3f} hbm {hbm:.3f} ms | "
              f"max diff {float((outs[0].float() - outs[1].float()).abs().max()):.3g}", flush=True)
        del ap, wp, r, out, outs
        torch.cuda.empty_cache()
    print(f"T={T} six GEMMs of a layer: old {tot[0]:.3f} ms, pp {tot[1]:.3f} ms, x{tot[0] / tot[1]:.3f}", flush=True)

# whole forward, bf16
from rnamsm.model import MSATransformer
state = synthetic.make_state_dict(seed=0)
m = MSATransformer(num_layers=10)
m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
m = m.eval().to(dev); m.gemm_dtype = "bf16"; m.check_finite = False
for (M, L) in ((256, 512), (1024, 1024)):
    toks = torch.from_numpy(synthetic.make_tokens(M, L, 0)).to(dev)
    res = {}
    for rnd in range(2):
        for v in (0, 1):
            setpp(v)
            o = m.forward_one(toks); torch.cuda.synchronize()
            res.setdefault(v, []).append(timeit(lambda: m.forward_one(toks), n=5))
            if rnd == 0:
                res[(v, "emb")] = o["emb"].clone()
    d = float((res[(0, "emb")] - res[(1, "emb")]).abs().max())
    print(f"forward bf16 M={M} L={L}: old {statistics.median(res[0]):.2f} ms, pp {statistics.median(res[1]):.2f} ms, x{statistics.median(res[0]) / statistics.median(res[1]):.3f}, emb max diff {d:.3g}", flush=True)
setpp(0)
