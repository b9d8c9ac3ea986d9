"""rnamsm -- MI355X-native implementation of RNA-MSM's axial-attention forward path.

Host side of the drop-in boundary: the reference's module / model / CLI interface for this one path,
executing on hand-written gfx950 HIP kernels behind the C ABI in include/rnamsm.h.
"""
import torch  # noqa: F401  -- must come first: torch has to bind ITS bundled HIP runtime before librnamsm_hip.so
#                      (linked against the system ROCm) is loaded, or torch.cuda stops seeing the device

from . import synthetic  # noqa: F401
from .alphabet import RNAAlphabet  # noqa: F401

__all__ = ["RNAAlphabet", "synthetic"]
