#!/usr/bin/env python3
"""Where the CLI's wall time goes on a list of tiny alignments (default, batched): cProfile of the main thread plus
accumulated time inside the reader's load_msa_tokens and the writer's np.save (they run on helper threads and share the GIL)."""
import cProfile, os, pstats, sys, tempfile, time, shutil, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import numpy as np, torch
from rnamsm import synthetic, inference
from rnamsm.config import Config
from rnamsm.model import MSATransformer
NS = int(os.environ.get("NSMALL", 64))
DLO, DHI = (int(v) for v in os.environ.get("DEPTH", "2,12").split(","))       # rows per alignment (inclusive); "small" list of cli_throughput.py: DEPTH=4,24 LEN=40,120
LLO, LHI = (int(v) for v in os.environ.get("LEN", "40,80").split(","))
state = synthetic.make_state_dict(seed=0)
model = MSATransformer(num_layers=10)
model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
letters = np.array(list("ACGU-"))
acc = {}
lock = threading.Lock()


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            with lock:
                c = acc.setdefault(name, [0, 0.0]); c[0] += 1; c[1] += time.perf_counter() - t0
    return w


inference.load_msa_tokens = timed("load_msa_tokens (reader thread)", inference.load_msa_tokens)
inference.np.save = timed("np.save (writer thread)", np.save)
for rnd in range(3):
    rng = np.random.RandomState(1)
    root = tempfile.mkdtemp(prefix="rnamsm_cli_small_", dir=os.environ.get("SCRATCH", "/tmp"))
    os.makedirs(os.path.join(root, "results"))
    ids = [f"small{i:03d}" for i in range(NS)]
    for i in ids:
        depth, length = int(rng.randint(DLO, DHI + 1)), int(rng.randint(LLO, LHI + 1))
        rows = letters[rng.randint(0, 5, size=(depth, length))]
        with open(os.path.join(root, "results", f"{i}.a2m_msa2"), "w") as f:
            for r in range(depth):
                f.write(f">s{r}\n{''.join(rows[r])}\n")
    open(os.path.join(root, "rna_id.txt"), "w").write("\n".join(ids) + "\n")
    cfg = Config()
    cfg.data.root_path, cfg.data.MSA_path, cfg.data.MSA_list = root, "results", "rna_id.txt"
    cfg.data.sample_method, cfg.data.max_seqs_per_msa = "first", int(os.environ.get("MAXSEQS", 64))
    acc.clear()
    prof = cProfile.Profile() if rnd == 2 else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if prof:
        prof.enable()
    inference.extract_feat(cfg, model=model)
    if prof:
        prof.disable()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"round {rnd}: {NS} tiny alignments in {dt * 1e3:.1f} ms; " + "; ".join(f"{k}: {v[0]} calls {v[1] * 1e3:.1f} ms" for k, v in acc.items()), flush=True)
    if prof:
        pstats.Stats(prof).sort_stats("cumulative").print_stats(45)
    shutil.rmtree(root)
