"""pytest config: registers the `gpu` marker and puts the product package and oracle on sys.path.

`-m "not gpu"` : oracle vs golden fixtures, host logic, C-ABI symbol check (no GPU needed).
`-m gpu`       : parity tests proper -- the HIP path through the C-ABI vs oracle / golden fixtures.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (os.path.join(ROOT, "rna-msm_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_sessionstart(session):
    """The CPU oracle (torch on the host) is the checker in many GPU tests.  On the GPU box's 256-thread host torch's default intra-op
    pool is 4-5x SLOWER than 32 threads on these sizes (bench.py's sweep: one layer 39 s at 256 threads, 8 s at 32) and its speed
    varies from box to box -- two of the round's suite runs took 575 / 689 s instead of 370.  32 threads for the whole session."""
    try:
        import torch
        torch.set_num_threads(min(32, os.cpu_count() or 1))
    except Exception:               # noqa: BLE001 -- a CPU-only collection without torch still has to work
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "slow: a GPU test of more than ~20 s whose claim a faster test of the default run also "
                                       "covers; deselected unless the -m expression names `slow` (-m \"gpu and slow\", or "
                                       "-m \"gpu or slow\" for everything) or RNAMSM_RUN_SLOW=1")


def pytest_collection_modifyitems(config, items):
    """The driver's `-m gpu` run has a wall-clock cap: tests marked `slow` stay out of it unless asked for."""
    expr = config.getoption("-m") or ""
    if "slow" in expr or os.environ.get("RNAMSM_RUN_SLOW") == "1":
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("slow") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


_LETTER = {4: "A", 5: "G", 6: "C", 7: "U", 8: "X", 10: "-"}


def write_full_2drb1_a2m(directory) -> str:
    """BASELINE configs[0]'s alignment at full depth as an .a2m_msa2 file, rebuilt from the token matrix the reference's
    reader produced for the shipped results/2DRB_1.a2m_msa2 (tokens_2DRB_1_full.npz: 1176 rows x 35 columns)."""
    toks = golden("tokens_2DRB_1_full.npz")["all_tokens"]
    path = os.path.join(str(directory), "2DRB_1.a2m_msa2")
    with open(path, "w") as f:
        for i, row in enumerate(toks):
            f.write(f">seq{i}\n" + "".join(_LETTER[int(t)] for t in row[1:]) + "\n")
    return path


@pytest.fixture(scope="session")
def full_2drb1_a2m(tmp_path_factory):
    return write_full_2drb1_a2m(tmp_path_factory.mktemp("a2m"))
