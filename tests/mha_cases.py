"""Inputs and weights of the generic-MHA fixture tests/golden/mha_general.npz, redrawn from the repo's counter-based generator
with the tags tests/golden/make_golden_r6.py used (the file itself holds only hand-built masks and the reference's outputs).
Shared by the oracle's CPU test and the HIP test."""
import numpy as np

from rnamsm import synthetic

SEED = 17
E, H = 128, 2


def rnd(tag, shape, std=1.0):
    return (std * synthetic.normal(f"mha6:{tag}", SEED, shape)).astype(np.float32)


def weights(tag, kdim=None, vdim=None, bias=True, bias_kv=False):
    """{parameter name: array} with the reference module's names (q_proj.weight ... out_proj.bias, bias_k, bias_v)."""
    shapes = {"q_proj.weight": (E, E), "k_proj.weight": (E, kdim or E), "v_proj.weight": (E, vdim or E), "out_proj.weight": (E, E)}
    if bias:
        shapes.update({f"{p}_proj.bias": (E,) for p in ("q", "k", "v", "out")})
    if bias_kv:
        shapes.update({"bias_k": (1, 1, E), "bias_v": (1, 1, E)})
    out = {}
    for name, shape in shapes.items():
        std = 0.05 if name.endswith("bias") else (0.3 if name.startswith("bias_") else 0.06)
        out[name] = rnd(f"{tag}.{name}", shape, std)
    return out
