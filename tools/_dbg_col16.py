import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import torch, numpy as np
from rnamsm import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(1)
def ref(eff, R, C, H, D):
    t = eff.double()
    q, k, v = 0.125 * t[:, :D].view(R, C, H, 64), t[:, D:2*D].view(R, C, H, 64), t[:, 2*D:].view(R, C, H, 64)
    p = torch.softmax(torch.einsum("ichd,jchd->hcij", q, k), -1)
    return torch.einsum("hcij,jchd->ichd", p, v).reshape(R * C, D)
for (split, fmt) in ((1, 0), (3, 0), (3, 1)):
    for R in (1, 7, 31, 32, 33, 34, 63, 64, 65, 96, 97, 130, 200, 257):
        C, H = 5, 2
        D = 64 * H
        qkv = torch.randn(R * C, 3 * D, device=dev)
        hi, lo = ops.split_bf16(qkv, want_lo=split == 3, fmt=fmt)
        ht = torch.float16 if fmt == 1 else torch.bfloat16
        eff = hi.view(ht).double() + (lo.view(ht).double() if lo is not None else 0)
        v = lambda a, b: (hi[:, a:b], None if lo is None else lo[:, a:b])
        want = ref(eff.cpu(), R, C, H, D)
        line = f"split {split} fmt {fmt} R {R:4d}:"
        for var in (1, 4, 5, 3):
            _lib.check(lib.rnamsm_set_param(b"attn16", var))
            got = ops.col_attn16(v(0, D), v(D, 2 * D), v(2 * D, 3 * D), R, C, H, fmt=fmt, scale=0.125).cpu().double()
            nbad = int((~torch.isfinite(got)).sum())
            err = float((got - want).norm() / want.norm()) if nbad == 0 else float("nan")
            line += f"  v{var} err {err:.2e} nonfinite {nbad}"
            if nbad:
                rows = torch.nonzero(~torch.isfinite(got).all(1)).flatten()[:6].tolist()
                line += f" rows {[(r // C, r % C) for r in rows]}"
        _lib.check(lib.rnamsm_set_param(b"attn16", 1))
        print(line, flush=True)
