import os, sys, time, statistics
sys.path.insert(0, "rna-msm_amd")
import torch
from rnamsm import synthetic
from rnamsm.model import MSATransformer
state = synthetic.make_state_dict(seed=0)
m = MSATransformer(num_layers=10)
m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
m = m.eval().to("cuda:0"); m.check_finite = False
for M, L, n in ((256, 300, 12), (512, 36, 24), (128, 256, 24)):
    toks = [torch.from_numpy(synthetic.make_tokens(M, L, i)).to("cuda:0") for i in range(n)]
    for t in toks[:2]: m.forward_one(t, need_repr=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in toks: m.forward_one(t, need_repr=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in toks: m.checked_forward_one(t, need_repr=False)
    torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    print(f"{M} x {L}: back-to-back outputs-only forwards {n * M * L / dt / 1e3:.1f} k residues/s ({1e3 * dt / n:.2f} ms each); with the per-MSA error-word sync {n * M * L / dt2 / 1e3:.1f} k ({1e3 * dt2 / n:.2f} ms)", flush=True)
