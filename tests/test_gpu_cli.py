"""End-to-end drop-in check on the GPU: the CLI (RNA_MSM_Inference.py, same overrides as the reference) turns a shipped
`.a2m_msa2` alignment + a Lightning-style checkpoint into `<id>_emb.npy` / `<id>_atp.npy` with the reference's
shape / dtype / layout contract, and the values match the oracle run on the same tokens."""
import os
import shutil
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, golden, rel_l2
from oracle import msm_oracle as O
from rnamsm import synthetic

pytestmark = pytest.mark.gpu


def test_cli_writes_reference_contract_files(tmp_path):
    sys.path.insert(0, ROOT)
    import RNA_MSM_Inference as cli
    state = synthetic.make_state_dict(seed=0)
    ckpt = tmp_path / "model.ckpt"
    torch.save({"state_dict": {k: torch.from_numpy(v) for k, v in state.items()}, "epoch": 0}, ckpt)   # Lightning layout
    msa_dir = tmp_path / "results"
    msa_dir.mkdir()
    shutil.copy(os.path.join(GOLDEN, "2DRB_1_first64.a2m_msa2"), msa_dir / "2DRB_1.a2m_msa2")
    (tmp_path / "rna_id.txt").write_text("2DRB_1\n")
    cli.main([f"data.root_path={tmp_path}", "data.MSA_path=results", f"data.model_path={ckpt}",
              "data.MSA_list=rna_id.txt", "data.max_seqs_per_msa=32", "data.sample_method=first"])
    emb = np.load(msa_dir / "2DRB_1_emb.npy")
    atp = np.load(msa_dir / "2DRB_1_atp.npy")
    assert emb.shape == (35, 768) and emb.dtype == np.float32 and emb.flags["C_CONTIGUOUS"]
    assert atp.shape == (120, 35, 35) and atp.dtype == np.float32 and atp.flags["C_CONTIGUOUS"]
    assert atp.sum(-1).max() <= 1.0 + 1e-5                      # <cls> column stripped -> rows sum to < 1
    toks = golden("tokens_2DRB_1_first64.npz")["tokens"][:32]
    res = O.forward(torch.from_numpy(toks), O.to_torch_params(state))
    o_emb, o_atp = O.pack_outputs(res)
    assert rel_l2(emb, o_emb) < 1e-4 and np.abs(atp - o_atp.numpy()).max() < 1e-4
    # strict checkpoint loading, as the reference (RNA_MSM_Inference.py:133-135)
    bad = {k: torch.from_numpy(v) for k, v in state.items() if k != "emb_layer_norm_after.bias"}
    torch.save({"state_dict": bad}, ckpt)
    with pytest.raises(RuntimeError):
        cli.main([f"data.root_path={tmp_path}", f"data.model_path={ckpt}", "data.sample_method=first"])


def test_async_io_pipeline_writes_the_same_files(tmp_path):
    """extract_feat(async_io=True): alignments are parsed one ahead on a helper thread and the outputs leave through a
    side-stream D2H copy + writer thread; the files must be byte-identical to the sequential loop's, in the same order."""
    from rnamsm.config import Config
    from rnamsm.inference import extract_feat
    from rnamsm.model import MSATransformer
    state = synthetic.make_state_dict(seed=0)
    model = MSATransformer(num_layers=10)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    ids = ["rnaC", "rnaA", "rnaD", "rnaB", "rnaE"]
    outs = {}
    for mode in (False, True):
        root = tmp_path / ("async" if mode else "sync")
        (root / "results").mkdir(parents=True)
        for n, i in enumerate(ids):           # different depths so consecutive MSAs differ in shape and content
            lines = open(os.path.join(GOLDEN, "2DRB_1_first64.a2m_msa2")).read().splitlines()
            (root / "results" / f"{i}.a2m_msa2").write_text("\n".join(lines[: 2 * (8 + 5 * n)]) + "\n")
        (root / "rna_id.txt").write_text("\n".join(ids) + "\n")
        cfg = Config()
        cfg.data.root_path, cfg.data.MSA_path, cfg.data.MSA_list = str(root), "results", "rna_id.txt"
        cfg.data.sample_method, cfg.data.max_seqs_per_msa = "first", 64
        written = extract_feat(cfg, model=model, async_io=mode)
        assert written == sorted(ids)
        outs[mode] = {f.name: f.read_bytes() for f in sorted((root / "results").glob("*.npy"))}
    assert len(outs[True]) == 2 * len(ids) and outs[True].keys() == outs[False].keys()
    for name in outs[True]:
        assert outs[True][name] == outs[False][name], name
    emb = np.load(tmp_path / "async" / "results" / "rnaA_emb.npy")
    assert emb.shape == (35, 768) and emb.dtype == np.float32


def _gather_worker(rank, world, port, root):
    """One rank of the CLI loop with gather_to_rank0 (both ranks on device 0, gloo: payloads staged through the host)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["LOCAL_RANK"] = "0"
    os.environ["RNAMSM_GATHER_BACKEND"] = "gloo"      # two ranks on ONE device: RCCL refuses that (tests/test_gpu_bench.py covers the refusal)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rnamsm.config import Config
        from rnamsm.inference import extract_feat
        from rnamsm.model import MSATransformer
        state = synthetic.make_state_dict(seed=0)
        model = MSATransformer(num_layers=10)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
        cfg = Config()
        cfg.data.root_path, cfg.data.MSA_path, cfg.data.MSA_list = str(root), "results", "rna_id.txt"
        cfg.data.sample_method, cfg.data.max_seqs_per_msa, cfg.data.device = "first", 64, "cuda:0"
        cfg.data.batch_small_msas = False     # byte-identity across DIFFERENT shardings needs the one-by-one loop (see below)
        written = extract_feat(cfg, model=model, gather_to_rank0=True)
        assert (len(written) == 5) if rank == 0 else (written == [])     # only rank 0 writes
    finally:
        dist.destroy_process_group()


def test_cli_gather_to_rank0_streams_rounds_and_writes_the_same_files(tmp_path):
    """with gather_to_rank0 the outputs travel to rank 0 one ROUND at a time (RoundGatherer) and leave through
    the async writer as they arrive -- five ids over two ranks (three rounds, the last with an item-less rank), files
    byte-identical to the single-process loop."""
    import socket
    import torch.multiprocessing as mp
    from rnamsm.config import Config
    from rnamsm.inference import extract_feat
    from rnamsm.model import MSATransformer
    ids = ["rnaC", "rnaA", "rnaD", "rnaB", "rnaE"]
    roots = {}
    for name in ("single", "gathered"):
        root = tmp_path / name
        (root / "results").mkdir(parents=True)
        lines = open(os.path.join(GOLDEN, "2DRB_1_first64.a2m_msa2")).read().splitlines()
        for n, i in enumerate(ids):
            (root / "results" / f"{i}.a2m_msa2").write_text("\n".join(lines[: 2 * (6 + 7 * n)]) + "\n")
        (root / "rna_id.txt").write_text("\n".join(ids) + "\n")
        roots[name] = root
    state = synthetic.make_state_dict(seed=0)
    model = MSATransformer(num_layers=10)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    cfg = Config()
    cfg.data.root_path, cfg.data.MSA_path, cfg.data.MSA_list = str(roots["single"]), "results", "rna_id.txt"
    cfg.data.sample_method, cfg.data.max_seqs_per_msa = "first", 64
    # the default CLI pads consecutive small alignments into one ragged batch; which alignments share a frame depends on the
    # sharding, and a frame's members agree with their lone forwards to fp32 rounding only -- the byte comparison below is
    # about the gather, so both sides run one by one
    cfg.data.batch_small_msas = False
    extract_feat(cfg, model=model)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_gather_worker, args=(2, port, roots["gathered"]), nprocs=2, join=True)
    a = {f.name: f.read_bytes() for f in sorted((roots["single"] / "results").glob("*.npy"))}
    b = {f.name: f.read_bytes() for f in sorted((roots["gathered"] / "results").glob("*.npy"))}
    assert len(a) == 10 and a.keys() == b.keys()
    for name in a:
        assert a[name] == b[name], name


@pytest.mark.parametrize("packing", [True, "split", False])
def test_cli_batches_small_alignments_by_default(tmp_path, packing, monkeypatch):
    """(packing = "split", round 6: the pool's token-packed groups are PIPELINED -- group g+1 is enqueued before group g's error word
    and outputs are read, on side streams behind an event -- and a pool that would be one large group is dealt into two; the split
    threshold is lowered to 1 token here so that this small list takes that path: same bytes.)
    data.batch_small_msas (default true; false = the reference's one-by-one loop): the small alignments of the id list
    (different depths and lengths here) share launch sets -- packing (data.pack_small_msas, the default): one token-packed group,
    nothing padded, the 4544-token alignment included; packing off: pooled, grouped by shape, padded into one frame per group, the
    4544-token one alone in between.  Same files, the returned ids in list order.  Packed (the default): every file is BYTE FOR BYTE
    the one-by-one run's (round 5: an id's files do not depend on what else is in the list, as in the reference's loop,
    RNA_MSM_Inference.py:141-166); framed (pack_small_msas=false): equal to fp32 rounding (masked attention over the frame), rnaD
    -- alone in between -- bit-identical."""
    from rnamsm.config import Config
    from rnamsm.inference import extract_feat
    from rnamsm.model import MSATransformer
    drop = ()
    if packing == "split":
        import rnamsm.inference as inf
        monkeypatch.setattr(inf, "PIPELINE_SPLIT_TOKENS", 1)
        packing, drop = True, ("rnaD",)     # seven small alignments of ONE fold class: one group in all -> dealt into two, pipelined
    state = synthetic.make_state_dict(seed=0)
    model = MSATransformer(num_layers=10)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    records = open(os.path.join(GOLDEN, "2DRB_1_first64.a2m_msa2")).read().splitlines()
    names, seqs = records[0::2], records[1::2]
    ids = [f"rna{c}" for c in "ABCDEFGH" if f"rna{c}" not in drop]
    shapes = {"rnaA": (3, 20), "rnaB": (9, 35), "rnaC": (5, 28), "rnaD": (64, 70), "rnaE": (12, 30), "rnaF": (2, 12),
              "rnaG": (7, 33), "rnaH": (1, 35)}                      # rnaD = 4544 tokens: not small, runs alone in between
    outs = {}
    for mode in (False, True):
        root = tmp_path / ("batched" if mode else "plain")
        (root / "results").mkdir(parents=True)
        for i in ids:
            depth, length = shapes[i]
            text = "".join(f"{names[r]}\n{(seqs[r] * 2)[:length]}\n" for r in range(depth))
            (root / "results" / f"{i}.a2m_msa2").write_text(text)
        (root / "rna_id.txt").write_text("\n".join(ids) + "\n")
        cfg = Config()
        cfg.data.root_path, cfg.data.MSA_path, cfg.data.MSA_list = str(root), "results", "rna_id.txt"
        cfg.data.sample_method, cfg.data.max_seqs_per_msa, cfg.data.batch_small_msas = "first", 64, mode
        cfg.data.pack_small_msas = packing
        assert extract_feat(cfg, model=model) == ids
        outs[mode] = {f.name: np.load(f) for f in sorted((root / "results").glob("*.npy"))}
    assert outs[True].keys() == outs[False].keys() and len(outs[True]) == 2 * len(ids)
    for name, want in outs[False].items():
        got = outs[True][name]
        assert got.shape == want.shape and got.dtype == want.dtype and got.flags["C_CONTIGUOUS"], name
        if packing:
            assert got.tobytes() == want.tobytes(), name
        elif name.endswith("_emb.npy"):
            assert rel_l2(got, want) < 1e-5, name
        else:
            assert np.abs(got - want).max() < 2e-5, name
    if not drop:
        assert np.array_equal(outs[True]["rnaD_emb.npy"], outs[False]["rnaD_emb.npy"])   # the large one ran alone: same bits


def test_cli_in_a_16bit_mode_packs_its_small_alignments_in_that_mode(tmp_path):
    """model.gemm_dtype=bf16 through the CLI (round 5): the alignments of <= 8192 tokens leave as ONE token-packed batch IN bf16
    (rnamsm_forward_packed with dtype bf16: Linear layers on the 16-bit matrix cores, attention on the exact kernels) -- their files
    sit at bf16's distance from the exact path's, as the one-by-one bf16 files do, and close to those; the larger ones run alone in
    bf16 (bit-identical to forward_one in that mode).  data.pack_small_msas=false keeps the one-by-one loop in the model's mode."""
    from rnamsm.config import Config
    from rnamsm.inference import extract_feat
    from rnamsm.model import MSATransformer
    state = synthetic.make_state_dict(seed=0)
    model = MSATransformer(num_layers=10)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    records = open(os.path.join(GOLDEN, "2DRB_1_first64.a2m_msa2")).read().splitlines()
    names, seqs = records[0::2], records[1::2]
    shapes = {"rnaA": (3, 20), "rnaB": (9, 35), "rnaC": (64, 140), "rnaD": (5, 28), "rnaE": (12, 30)}     # rnaC: 9024 tokens > 8192
    ids = sorted(shapes)

    def run(tag, dtype, **flags):
        root = tmp_path / tag
        (root / "results").mkdir(parents=True)
        for i in ids:
            depth, length = shapes[i]
            (root / "results" / f"{i}.a2m_msa2").write_text("".join(f"{names[r]}\n{(seqs[r] * 5)[:length]}\n" for r in range(depth)))
        (root / "rna_id.txt").write_text("\n".join(ids) + "\n")
        cfg = Config()
        cfg.data.root_path, cfg.data.MSA_path, cfg.data.MSA_list = str(root), "results", "rna_id.txt"
        cfg.data.sample_method, cfg.data.max_seqs_per_msa = "first", 64
        for k, v in flags.items():
            setattr(cfg.data, k, v)
        cfg.model.gemm_dtype = dtype                     # (the CLI's override: model.gemm_dtype=bf16)
        try:
            assert extract_feat(cfg, model=model) == ids
        finally:
            model.gemm_dtype = "f32"
        return {f.name: np.load(f) for f in sorted((root / "results").glob("*.npy"))}
    exact = run("exact_one_by_one", "f32", batch_small_msas=False)
    bf16_alone = run("bf16_one_by_one", "bf16", batch_small_msas=False)
    default = run("bf16_default", "bf16")
    own = run("bf16_own_arithmetic", "bf16", pack_small_msas=False)
    for i in ids:
        e = f"{i}_emb.npy"
        if i == "rnaC":
            assert np.array_equal(default[e], bf16_alone[e])                       # above the limit: alone, in the model's mode
        else:
            d_pk, d_one = rel_l2(default[e], exact[e]), rel_l2(bf16_alone[e], exact[e])
            assert 1e-4 < d_pk < 5e-2 and 1e-4 < d_one < 5e-2, (i, d_pk, d_one)    # both carry bf16's rounding, neither is the exact path
            assert d_pk < 2.0 * d_one + 1e-3, (i, d_pk, d_one)                     # ... the packed batch no further from it than alone
        assert np.array_equal(own[e], bf16_alone[e]), i


def test_cli_in_a_16bit_mode_a_lone_small_alignment_gets_the_arithmetic_of_a_group(tmp_path):
    """what a small alignment's files hold must not depend on whether it had company.  A bf16 list with exactly ONE
    small alignment (and one large one, which runs alone in bf16): the small one runs as a packed batch of one -- the arithmetic
    of a group (16-bit Linear layers, exact attention) -- so its files equal what it gets inside a group of small ones (same
    GEMM kernel at these sizes: the same bits), and carry bf16's distance from the exact path either way."""
    from rnamsm.config import Config
    from rnamsm.inference import extract_feat
    from rnamsm.model import MSATransformer
    state = synthetic.make_state_dict(seed=0)
    model = MSATransformer(num_layers=10)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    records = open(os.path.join(GOLDEN, "2DRB_1_first64.a2m_msa2")).read().splitlines()
    names, seqs = records[0::2], records[1::2]
    shapes = {"rnaA": (9, 35), "rnaB": (64, 140), "rnaC": (3, 20), "rnaD": (5, 28)}      # rnaB: 9024 tokens > 8192

    def run(tag, dtype, ids, **flags):
        root = tmp_path / tag
        (root / "results").mkdir(parents=True)
        for i in ids:
            depth, length = shapes[i]
            (root / "results" / f"{i}.a2m_msa2").write_text("".join(f"{names[r]}\n{(seqs[r] * 5)[:length]}\n" for r in range(depth)))
        (root / "rna_id.txt").write_text("\n".join(ids) + "\n")
        cfg = Config()
        cfg.data.root_path, cfg.data.MSA_path, cfg.data.MSA_list = str(root), "results", "rna_id.txt"
        cfg.data.sample_method, cfg.data.max_seqs_per_msa = "first", 64
        for k, v in flags.items():
            setattr(cfg.data, k, v)
        cfg.model.gemm_dtype = dtype
        try:
            assert extract_feat(cfg, model=model) == sorted(ids)
        finally:
            model.gemm_dtype = "f32"
        return {f.name: np.load(f) for f in sorted((root / "results").glob("*.npy"))}
    exact = run("exact", "f32", ["rnaA"], batch_small_msas=False)
    lone = run("bf16_lone_small", "bf16", ["rnaA", "rnaB"])
    grouped = run("bf16_group_of_small", "bf16", ["rnaA", "rnaB", "rnaC", "rnaD"])
    for name in ("rnaA_emb.npy", "rnaA_atp.npy"):
        assert np.array_equal(lone[name], grouped[name]), name                    # the same bits with or without company
    assert 1e-4 < rel_l2(lone["rnaA_emb.npy"], exact["rnaA_emb.npy"]) < 5e-2      # bf16's rounding: the mode that was asked for
    assert np.array_equal(lone["rnaB_emb.npy"], grouped["rnaB_emb.npy"])          # the large one: alone in bf16 either way


@pytest.mark.parametrize("batching", [True, False])
def test_a_bad_alignment_late_in_the_list_does_not_cost_the_results_before_it(tmp_path, batching):
    """The CLI loop is pipelined (round 6): a forward is enqueued before the previous one's results are read back, small alignments
    wait in a pool.  An alignment that cannot be read (an invalid character: the reference raises ValueError("Invalid tokens in
    input"), utils/tokenization.py:107-129) must still surface as that error -- and everything computed or read BEFORE it must be on
    disk, byte for byte what a clean run over those ids writes (the one-by-one loop of the reference would have written them before
    reaching the bad file).  batching = the pooled / packed default and the strictly one-by-one loop."""
    from rnamsm.config import Config
    from rnamsm.inference import extract_feat
    from rnamsm.model import MSATransformer
    state = synthetic.make_state_dict(seed=0)
    model = MSATransformer(num_layers=10)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    records = open(os.path.join(GOLDEN, "2DRB_1_first64.a2m_msa2")).read().splitlines()
    names, seqs = records[0::2], records[1::2]
    shapes = {"rnaA": (5, 30), "rnaB": (9, 35), "rnaC": (3, 22), "rnaD": (4, 25), "rnaE": (6, 28)}
    outs = {}
    for kind, ids in (("clean", ["rnaA", "rnaB", "rnaC"]), ("broken", ["rnaA", "rnaB", "rnaC", "rnaD", "rnaE"])):
        root = tmp_path / kind
        (root / "results").mkdir(parents=True)
        for i in ids:
            depth, length = shapes[i]
            text = "".join(f"{names[r]}\n{(seqs[r] * 2)[:length]}\n" for r in range(depth))
            if i == "rnaD":
                text = text.replace("A", "!", 1)                    # not a residue, not an insertion marker
            (root / "results" / f"{i}.a2m_msa2").write_text(text)
        (root / "rna_id.txt").write_text("\n".join(ids) + "\n")
        cfg = Config()
        cfg.data.root_path, cfg.data.MSA_path, cfg.data.MSA_list = str(root), "results", "rna_id.txt"
        cfg.data.sample_method, cfg.data.max_seqs_per_msa, cfg.data.batch_small_msas = "first", 64, batching
        if kind == "clean":
            assert extract_feat(cfg, model=model) == ids
        else:
            with pytest.raises(ValueError, match="Invalid tokens"):
                extract_feat(cfg, model=model)
        outs[kind] = {f.name: f.read_bytes() for f in sorted((root / "results").glob("*.npy"))}
    assert len(outs["clean"]) == 6
    for name, want in outs["clean"].items():
        assert outs["broken"].get(name) == want, name                # everything before the bad file: written, same bytes
    assert not any(n.startswith(("rnaD", "rnaE")) for n in outs["broken"])
