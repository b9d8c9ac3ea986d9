"""ctypes binding of librnamsm_hip.so (include/rnamsm.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  `load()` raises if the
shared object is missing, and every op wrapper raises if its tensors are not on a HIP device.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (load order: see rnamsm/__init__.py)
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RNAMSM_LIB_PATH") or os.path.join(_HERE, "librnamsm_hip.so")    # override: A/B builds only

RNAMSM_OK = 0
F32, BF16, F16X3 = 0, 1, 3        # (2 was "bf16x3", removed in round 5: include/rnamsm.h)
DTYPES = {"f32": F32, "bf16": BF16, "f16x3": F16X3}
ACT_NONE, ACT_GELU_ERF = 0, 1
OUT_REPR = 1

# index tables of rnamsm_forward's weight-pointer array (include/rnamsm.h)
W_GLOBAL = ("embed_tokens", "embed_positions", "row_pos", "ln_before_g", "ln_before_b", "ln_after_g", "ln_after_b")
W_LAYER = ("row_ln_g", "row_ln_b", "row_wqkv", "row_bqkv", "row_wo", "row_bo",
           "col_ln_g", "col_ln_b", "col_wqkv", "col_bqkv", "col_wo", "col_bo",
           "ffn_ln_g", "ffn_ln_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")


class ModelDims(ctypes.Structure):
    _fields_ = [("num_layers", c_int), ("embed_dim", c_int), ("num_heads", c_int), ("ffn_dim", c_int),
                ("vocab", c_int), ("num_positions", c_int), ("pad_idx", c_int), ("ln_eps", c_float),
                ("row_pos_dim", c_int)]          # 0 / 1: scalar per alignment row; embed_dim: the msm/ variant's per-channel rows


_SIGNATURES = {
    "rnamsm_version": (c_int, []),
    "rnamsm_last_error": (c_char_p, []),
    "rnamsm_device_count": (c_int, []),
    "rnamsm_embed_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "rnamsm_embed_ln_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                     c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "rnamsm_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "rnamsm_gemm_bias_act_res": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                         c_int64, c_int, c_int, c_int, c_float, c_int, c_void_p, c_int, c_void_p]),
    "rnamsm_gemm_row_scaled": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_float,
                                       c_int, c_void_p, c_int, c_void_p]),
    "rnamsm_split_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "rnamsm_gemm_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                 c_int64, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p]),
    "rnamsm_layernorm_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int,
                                       c_void_p]),
    "rnamsm_row_logits_nsplit": (c_int, [c_int, c_int, c_int]),
    "rnamsm_row_logits_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "rnamsm_row_logits": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rnamsm_softmax_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "rnamsm_softmax_rows_scaled": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_float, c_void_p]),
    "rnamsm_row_chunks": (c_int, [c_int, c_int, c_int]),
    "rnamsm_row_logits_chunked": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_void_p]),
    "rnamsm_softmax_rows_chunked": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "rnamsm_row_apply": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "rnamsm_col_attn_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int,
                                      c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "rnamsm_row_logits16_nsplit": (c_int, [c_int, c_int, c_int, c_int]),
    "rnamsm_row_logits16_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "rnamsm_row_logits16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int,
                                    c_float, c_int, c_void_p]),
    "rnamsm_softmax_rows_planes": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_int, c_int,
                                           c_void_p, c_int, c_void_p]),
    "rnamsm_row_apply16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int,
                                   c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p]),
    "rnamsm_col_attn16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                  c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "rnamsm_col_attn16_prescaled": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                            c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rnamsm_zero_plane_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p]),
    "rnamsm_add": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "rnamsm_head_mean": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "rnamsm_pad_mask": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "rnamsm_pack_outputs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rnamsm_contact_head_workspace_bytes": (c_size_t, [c_int, c_int]),
    "rnamsm_contact_head": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "rnamsm_greedy_select_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "rnamsm_greedy_select": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rnamsm_msa_weights": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p]),
    "rnamsm_forward_workspace_bytes": (c_size_t, [POINTER(ModelDims), c_int, c_int, c_int, c_int]),
    "rnamsm_forward": (c_int, [POINTER(ModelDims), POINTER(c_void_p), c_void_p, c_int, c_int, c_void_p, c_size_t,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                               POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_void_p]),
    "rnamsm_forward_batch_workspace_bytes": (c_size_t, [POINTER(ModelDims), c_int, c_int, c_int]),
    "rnamsm_forward_batch": (c_int, [POINTER(ModelDims), POINTER(c_void_p), c_void_p, c_int, c_int, c_int, c_void_p, c_size_t,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, POINTER(c_void_p), c_int,
                                     POINTER(c_void_p), c_void_p]),
    "rnamsm_forward_packed_workspace_bytes": (c_size_t, [POINTER(ModelDims), c_int, c_void_p]),
    "rnamsm_forward_packed": (c_int, [POINTER(ModelDims), POINTER(c_void_p), c_void_p, c_int, c_void_p, c_void_p, c_size_t,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_void_p), c_int, POINTER(c_void_p),
                                      c_void_p]),
    "rnamsm_ln_fold_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                       c_void_p]),
    "rnamsm_row_partials": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "rnamsm_gemm_residual_stats": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                           c_int64, c_int, c_int, c_void_p, c_int64, c_int, c_void_p]),
    "rnamsm_gemm16_lnfold": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_void_p]),
    "rnamsm_gemm16_residual_stats": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                             c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p,
                                             c_int64, c_void_p]),
    "rnamsm_row_stats_from_partials": (c_int, [c_void_p, c_int64, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "rnamsm_gemm_lnfold": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                   c_int64, c_int64, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "rnamsm_col_attn_fused_prescaled": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int,
                                                c_int, c_int, c_void_p]),
    "rnamsm_col_attn_fused_queries": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int,
                                              c_int, c_int, c_void_p, c_int, c_void_p]),
    "rnamsm_col_attn_probs": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_float,
                                      c_int, c_void_p]),
    "rnamsm_col_attn_probs16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int,
                                        c_void_p, c_int, c_float, c_void_p]),
    "rnamsm_timing_enable": (c_int, [c_int]),
    "rnamsm_timing_collect": (c_int, []),
    "rnamsm_timing_get": (c_int, [c_int, POINTER(c_char_p), POINTER(ctypes.c_longlong), POINTER(ctypes.c_double),
                                  POINTER(ctypes.c_double), POINTER(ctypes.c_double)]),
    "rnamsm_timing_get_bound": (c_int, [c_int, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(ctypes.c_double)]),
    "rnamsm_timing_get_valu_bound": (c_int, [c_int, POINTER(ctypes.c_double)]),
    "rnamsm_timing_reset": (None, []),
    "rnamsm_set_param": (c_int, [c_char_p, c_int]),
    "rnamsm_get_param": (c_int, [c_char_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class RnamsmError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the HIP library; loud failure if it has not been built (python __graft_entry__.py / make -C csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RnamsmError(
                f"{LIB_PATH} not found: the HIP extension is the only implementation of this path "
                "(no CPU fallback). Build it with `make -C rna-msm_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if the header and the library disagree
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != RNAMSM_OK:
        msg = load().rnamsm_last_error().decode(errors="replace")
        if rc == -2:
            raise NotImplementedError(msg)
        raise RnamsmError(f"librnamsm_hip error {rc}: {msg}")


def kernel_timings() -> dict:
    """Fold completed HIP-event pairs and return {kernel: {launches, ms, flops, bytes, bound_ms, mfma_bound_ms, hbm_bound_ms,
    valu_bound_ms}} (rnamsm_timing_*; bound_ms = per-launch max(matrix, HBM, vector-ALU) roofline times summed over the launches)."""
    lib = load()
    n = lib.rnamsm_timing_collect()
    out = {}
    for c in range(n):
        name, cnt = c_char_p(), ctypes.c_longlong()
        ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        check(lib.rnamsm_timing_get(c, ctypes.byref(name), ctypes.byref(cnt), ctypes.byref(ms), ctypes.byref(fl),
                                    ctypes.byref(by)))
        bd, mf, hb = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        check(lib.rnamsm_timing_get_bound(c, ctypes.byref(bd), ctypes.byref(mf), ctypes.byref(hb)))
        va = ctypes.c_double()
        check(lib.rnamsm_timing_get_valu_bound(c, ctypes.byref(va)))
        out[name.value.decode()] = {"launches": cnt.value, "ms": ms.value, "flops": fl.value, "bytes": by.value,
                                    "bound_ms": bd.value, "mfma_bound_ms": mf.value, "hbm_bound_ms": hb.value,
                                    "valu_bound_ms": va.value}
    return out
