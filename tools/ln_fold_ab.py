#!/usr/bin/env python3
"""A/B of the folded LayerNorm (knob "ln_fold") on the exact path, one process: ms per forward with LayerNorm applied
inside the QKV / fc1 GEMMs vs separate LayerNorm launches, plus the per-category kernel times of each variant.

    python tools/ln_fold_ab.py [M L [reps]]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))

import torch

from rnamsm import _lib, ops, synthetic
from rnamsm.model import MSATransformer

M, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 512)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ON = int(os.environ.get("LN_FOLD_ON", 3))          # 3 = folded at every shape (the default knob value 1 folds from 16384 tokens)
dev = torch.device("cuda:0")
model = MSATransformer(num_layers=10)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(seed=0).items()}, strict=True)
model = model.eval().to(dev)
model.gemm_dtype = os.environ.get("DTYPE", "f32")      # f32 | f16x3 | bf16
model.check_finite = False
toks = torch.from_numpy(synthetic.make_tokens(M, L, 0)).to(dev)
lib = _lib.load()


def timed(fold, need_repr=True):
    ops.set_param("ln_fold", fold)
    for _ in range(2):
        model.forward_one(toks, has_padding=False, need_repr=need_repr)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = model.forward_one(toks, has_padding=False, need_repr=need_repr)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    lib.rnamsm_timing_reset()
    lib.rnamsm_timing_enable(1)
    model.forward_one(toks, has_padding=False, need_repr=need_repr)
    torch.cuda.synchronize()
    lib.rnamsm_timing_enable(0)
    cats = {}
    import ctypes
    for c in range(lib.rnamsm_timing_collect()):
        name, n, t, fl, by = ctypes.c_char_p(), ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        lib.rnamsm_timing_get(c, ctypes.byref(name), ctypes.byref(n), ctypes.byref(t), ctypes.byref(fl), ctypes.byref(by))
        if n.value:
            cats[name.value.decode()] = {"launches": n.value, "ms": round(t.value, 3)}
    return ms, cats, out


res = {}
outs = {}
for fold in (0, ON, 0, ON):
    ms, cats, out = timed(fold)
    res.setdefault(f"ln_fold={fold}", []).append(round(ms, 3))
    res[f"kernels ln_fold={fold}"] = cats
    outs[fold] = out
for fold in (0, ON):
    ms, _, _ = timed(fold, need_repr=False)
    res[f"outputs-only ln_fold={fold}"] = round(ms, 3)
ops.set_param("ln_fold", 1)
d = (outs[ON]["emb"] - outs[0]["emb"]).double()
res["emb rel-L2 fold vs separate"] = float(d.norm() / outs[0]["emb"].double().norm())
res["atp max-abs fold vs separate"] = float((outs[ON]["atp"] - outs[0]["atp"]).abs().max())
res["shape"] = [M, L]
res["gemm_dtype"] = model.gemm_dtype
print(json.dumps(res, indent=1))
