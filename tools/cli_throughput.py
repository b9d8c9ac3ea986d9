#!/usr/bin/env python3
"""Wall-clock throughput of the CLI loop (rnamsm.inference.extract_feat) on synthetic alignments, sequential vs
pipelined I/O: N alignments of M sequences x L columns written to a scratch directory, forward in exact fp32."""
import os, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rna-msm_amd"))
import numpy as np, torch
from rnamsm import synthetic
from rnamsm.config import Config
from rnamsm import inference
from rnamsm.inference import extract_feat
if os.environ.get("SMALL_TOKENS"):                       # A/B of the "small alignment" limit
    inference.SMALL_MSA_TOKENS = int(os.environ["SMALL_TOKENS"])
if os.environ.get("PACKED_TOKENS"):                      # A/B of the token budget of a token-packed group
    inference.PACKED_TOKENS = int(os.environ["PACKED_TOKENS"])
from rnamsm.model import MSATransformer
N, M, L = int(os.environ.get("N", 12)), int(os.environ.get("M", 256)), int(os.environ.get("L", 300))
state = synthetic.make_state_dict(seed=0)
model = MSATransformer(num_layers=10)
model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
MODE = os.environ.get("MODE", "f32")                      # MODE=bf16 | f16x3: the CLI in a 16-bit arithmetic mode (model.gemm_dtype=...)
if os.environ.get("PACKED_SMALL"):                       # A/B of the size up to which an alignment waits for company (exact mode)
    inference.PACKED_SMALL_TOKENS = int(os.environ["PACKED_SMALL"])
if os.environ.get("PACKED_SMALL_16"):
    inference.PACKED_SMALL_TOKENS_16BIT = {k: int(os.environ["PACKED_SMALL_16"]) for k in inference.PACKED_SMALL_TOKENS_16BIT}
rng = np.random.RandomState(0)
letters = np.array(list("ACGU-"))
for mode in (False, True, False, True):
    root = tempfile.mkdtemp(prefix="rnamsm_cli_", dir=os.environ.get("SCRATCH", "/tmp"))
    os.makedirs(os.path.join(root, "results"))
    ids = [f"rna{i:03d}" for i in range(N)]
    for i in ids:
        rows = letters[rng.randint(0, 5, size=(M, L))]
        with open(os.path.join(root, "results", f"{i}.a2m_msa2"), "w") as f:
            for r in range(M):
                f.write(f">s{r}\n{''.join(rows[r])}\n")
    open(os.path.join(root, "rna_id.txt"), "w").write("\n".join(ids) + "\n")
    cfg = Config()
    cfg.model.gemm_dtype = MODE
    cfg.data.root_path, cfg.data.MSA_path, cfg.data.MSA_list = root, "results", "rna_id.txt"
    cfg.data.sample_method, cfg.data.max_seqs_per_msa = "first", M
    torch.cuda.synchronize(); t0 = time.perf_counter()
    extract_feat(cfg, model=model, async_io=mode)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nbytes = sum(os.path.getsize(os.path.join(root, "results", f)) for f in os.listdir(os.path.join(root, "results")) if f.endswith(".npy"))
    print(f"async_io={mode}: {N} MSAs ({M} x {L}) in {dt:.2f} s = {N / dt:.2f} MSA/s, {N * M * L / dt:.0f} residues/s, {nbytes / 1e6:.0f} MB written", flush=True)
    shutil.rmtree(root)

# ---- a list of SMALL alignments: the default CLI (data.batch_small_msas=true: consecutive small alignments
# share one ragged launch set) against the strictly one-by-one loop of the reference (data.batch_small_msas=false)
NS = int(os.environ.get("NSMALL", 64))
# two populations: "tiny" (2-12 rows x 40-80 columns) and "small" (4-24 rows x 40-120 columns: up to 2.9 k tokens each)
for label, (dlo, dhi), (llo, lhi) in (("tiny", (2, 13), (40, 81)), ("small", (4, 25), (40, 121)), ("mid", (16, 65), (60, 201))):
    if not NS:
        break
    res = {}
    for rnd in range(2):
        for batching in (False, True, "framed"):          # "framed": batched, but padded frames instead of token-packed groups (round 3)
            rng = np.random.RandomState(1)
            root = tempfile.mkdtemp(prefix="rnamsm_cli_small_", dir=os.environ.get("SCRATCH", "/tmp"))
            os.makedirs(os.path.join(root, "results"))
            ids = [f"small{i:03d}" for i in range(NS)]
            tokens = 0
            for i in ids:
                depth, length = int(rng.randint(dlo, dhi)), int(rng.randint(llo, lhi))
                tokens += depth * (length + 1)
                rows = letters[rng.randint(0, 5, size=(depth, length))]
                with open(os.path.join(root, "results", f"{i}.a2m_msa2"), "w") as f:
                    for r in range(depth):
                        f.write(f">s{r}\n{''.join(rows[r])}\n")
            open(os.path.join(root, "rna_id.txt"), "w").write("\n".join(ids) + "\n")
            cfg = Config()
            cfg.model.gemm_dtype = MODE
            cfg.data.root_path, cfg.data.MSA_path, cfg.data.MSA_list = root, "results", "rna_id.txt"
            cfg.data.sample_method, cfg.data.max_seqs_per_msa, cfg.data.batch_small_msas = "first", 64, bool(batching)
            cfg.data.pack_small_msas = batching is True
            torch.cuda.synchronize(); t0 = time.perf_counter()
            extract_feat(cfg, model=model)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            res.setdefault(batching, []).append(dt)
            shutil.rmtree(root)
    one, bat, fr = min(res[False]), min(res[True]), min(res["framed"])
    print(f"{NS} {label} alignments ({dlo}-{dhi - 1} rows x {llo}-{lhi - 1} columns, {tokens} tokens): one by one {one:.3f} s = {NS / one:.1f} MSA/s; "
          f"default (token-packed groups) {bat:.3f} s = {NS / bat:.1f} MSA/s; x{one / bat:.2f}; framed groups (round 3) {fr:.3f} s; x{one / fr:.2f}", flush=True)
