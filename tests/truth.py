"""Full-size truth for the parity tests: the ORACLE's code (oracle/msm_oracle.py) run on the GPU box's device through
torch -- float64 as the truth, float32 / bfloat16 as "what the reference's own arithmetic does at this size".

Test infrastructure only (it imports oracle/): the product never sees this file.  The CPU oracle needs ~5 minutes for one
M=256 x L=512 forward and cannot do M=L=1024 at all inside the suite's budget, so at BASELINE's full sizes the same
restatement is evaluated by torch's fp64 kernels on the device (a few seconds).  tests/test_gpu_fullsize.py first ties
this evaluation to the reference's own fp64 fixture (forward_m16_c33_fp64.npz) and to the CPU oracle's fp32 noise.
"""
import functools

import numpy as np
import torch

from oracle import msm_oracle as O
from rnamsm import synthetic

_PARAMS = {}


@functools.lru_cache(maxsize=None)
def state():
    return synthetic.make_state_dict(seed=0)


def params(dtype, device):
    key = (dtype, str(device))
    if key not in _PARAMS:
        _PARAMS[key] = O.to_torch_params(state(), dtype, device)
    return _PARAMS[key]


def oracle_outputs(tokens: np.ndarray, dtype, device, max_tokens=None):
    """(emb [L,768], atp [120,L,L]) of the oracle in `dtype` on `device`, returned as float64 tensors on that device."""
    torch.backends.cuda.matmul.allow_tf32 = False
    with torch.no_grad():
        res = O.forward(torch.from_numpy(tokens).to(device), params(dtype, device), max_tokens=max_tokens,
                        ffn_token_chunk=32768)
        emb, atp = O.pack_outputs(res)
        emb, atp = emb.double(), atp.double()
    del res
    if torch.device(device).type == "cuda":
        torch.cuda.empty_cache()
    return emb, atp


def errors(emb, atp, t_emb, t_atp) -> dict:
    """Deviation of (emb, atp) from the truth (t_emb, t_atp); everything in float64 on the truth's device."""
    emb = emb.to(t_emb.device, torch.float64)
    atp = atp.to(t_atp.device, torch.float64)
    de, da = emb - t_emb, atp - t_atp
    return {"emb_rel_l2": float(de.norm() / t_emb.norm()),
            "emb_max_abs_over_max": float(de.abs().max() / t_emb.abs().max()),
            "atp_rel_l2": float(da.norm() / t_atp.norm()),
            "atp_max_abs": float(da.abs().max()),
            "atp_mean_abs": float(da.abs().mean())}
