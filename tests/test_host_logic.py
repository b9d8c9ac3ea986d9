"""CPU tests of the host side: tokenizer / a2m reader (bit-exact vs reference fixtures), config overrides, CLI helper
behaviour, the C-ABI library (loads, exports every symbol of include/rnamsm.h, validates arguments without a GPU),
and the no-fallback rule (ops refuse CPU tensors)."""
import json
import os
import sys
import re
import subprocess

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, golden
from rnamsm import msa, synthetic
from rnamsm.alphabet import RNAAlphabet
from rnamsm.config import Config, parse_overrides
from oracle import tokenizer_oracle as TO


@pytest.fixture(scope="module")
def lib():
    from rnamsm import _lib
    if not os.path.exists(_lib.LIB_PATH):      # fresh checkout: hipcc cross-compiles gfx950 without a GPU
        subprocess.run(["make", "-C", os.path.join(ROOT, "rna-msm_amd", "csrc"), "-j8"], check=True)
    return _lib.load()


# ------------------------------------------------------------------ tokenizer / reader (integer path: bit-exact)
def test_alphabet_matches_reference_vocab():
    meta = json.load(open(os.path.join(GOLDEN, "tokens_meta.json")))
    a = RNAAlphabet.from_architecture("rna language")
    assert a.to_dict() == meta["vocab"] and a.prepend_bos == meta["prepend_bos"] and a.append_eos == meta["append_eos"]
    assert len(a) == 12 and a.pad_idx == 1 and a.cls_idx == 0 and a.mask_idx == 11
    with pytest.raises(ValueError):
        RNAAlphabet.from_architecture("ESM-1b")


@pytest.mark.parametrize("a2m,fixture", [("2DRB_1_first64.a2m_msa2", "tokens_2DRB_1_first64.npz"),
                                         ("synthetic_chars.a2m_msa2", "tokens_synthetic_chars.npz")])
def test_reader_tokens_bit_exact(a2m, fixture):
    toks = msa.load_msa_tokens(os.path.join(GOLDEN, a2m), RNAAlphabet())
    want = golden(fixture)["tokens"]
    assert toks.dtype == np.int64 and np.array_equal(toks, want)


def test_reader_agrees_with_oracle_tokenizer_on_random_alignments():
    from oracle import tokenizer_oracle as TO
    rng = np.random.RandomState(0)
    chars = np.array(list("ACGUTXRYKMSWBDHVN-acgu.*"))
    a = RNAAlphabet()
    for trial in range(20):
        R, L = rng.randint(1, 9), rng.randint(1, 40)
        rows = []
        for _ in range(R):
            body = list(rng.choice(list("ACGUTXRYKMSWBDHVN-"), L))
            for _ in range(rng.randint(0, 5)):                     # insertions are dropped, so they keep rows aligned
                body.insert(rng.randint(0, len(body) + 1), rng.choice(list("acgun.*")))
            rows.append("".join(body))
        text = "".join(f">s{i} x\n{r}\n" for i, r in enumerate(rows))
        got = a.encode_a2m_records([s for _, s in msa.read_fasta_records(text, is_text=True)])
        assert np.array_equal(got, TO.encode_msa(text))


def test_reader_errors_match_reference():
    a = RNAAlphabet()
    with pytest.raises(AssertionError, match="Seqlen Mismatch"):
        a.encode_a2m_records(["ACGU", "ACG"])
    with pytest.raises(ValueError, match="Invalid tokens in input"):
        a.encode_a2m_records(["ACGE", "ACGU"])
    with pytest.raises(ValueError, match="Invalid tokens in input"):
        a.encode(["ACGT"])                       # encode() takes already-clean sequences: T is not in the alphabet
    assert np.array_equal(a.encode("ACGU-XN"), np.array([[0, 4, 6, 5, 7, 10, 8, 9]]))


def test_row_subsampling_matches_reference_greedy_select():
    g = golden("tokens_2DRB_1_first64.npz")
    path = os.path.join(GOLDEN, "2DRB_1_first64.a2m_msa2")
    a = RNAAlphabet()
    assert np.array_equal(msa.load_msa_tokens(path, a, 16, "diversity-max"), g["diversity_max_16"])
    assert np.array_equal(msa.load_msa_tokens(path, a, 16, "diversity-min"), g["diversity_min_16"])
    assert np.array_equal(msa.load_msa_tokens(path, a, 16, "first"), g["tokens"][:16])
    assert np.array_equal(msa.load_msa_tokens(path, a, 64, "hhfilter"), g["tokens"])      # no sub-sampling needed
    with pytest.raises(NotImplementedError):
        msa.load_msa_tokens(path, a, 16, "hhfilter")
    with pytest.raises(AssertionError):
        msa.load_msa_tokens(path, a, 16, "nonsense")


def test_sample_pretrained_subsampling_matches_reference_weights_and_draws(full_2drb1_a2m):
    """utils/align.py:150-163, 250-253 on the shipped 1176-row alignment: the sequence weights are the reference's bit for
    bit, and with numpy's generator in the same state (the reference draws from the global one, seeded by
    seed_everything) the weighted draw picks the reference's rows."""
    g = golden("msa_weights_2DRB_1.npz")
    path = full_2drb1_a2m
    a = RNAAlphabet()
    toks = msa.load_msa_tokens(path, a, None)
    assert np.array_equal(msa.msa_weights(toks, float(g["seqid_cutoff"])), g["weights"])
    for seed, n in ((42, 512), (7, 64)):
        got = msa.load_msa_tokens(path, a, n, "sample-pretrained", rng=np.random.RandomState(seed))
        assert np.array_equal(got, g[f"tokens_seed{seed}_n{n}"]) and (got[0] == toks[0]).all()
    np.random.seed(7)                                                   # rng=None draws from numpy's global state
    assert np.array_equal(msa.load_msa_tokens(path, a, 64, "sample-pretrained"), g["tokens_seed7_n64"])
    assert np.array_equal(msa.sample_weights(toks[:10], 10), np.arange(10))


def test_synthetic_generators_are_pure_functions():
    t1, t2 = synthetic.make_tokens(7, 19, 3), synthetic.make_tokens(7, 19, 3)
    assert np.array_equal(t1, t2) and (t1[:, 0] == 0).all() and set(np.unique(t1[:, 1:])) <= {4, 5, 6, 7, 8, 10}
    assert not np.array_equal(t1, synthetic.make_tokens(7, 19, 4))
    sd = synthetic.make_state_dict(seed=1, embed_dim=128, num_layers=2, num_heads=2)
    b = sd["layers.0.row_self_attention.layer.q_proj.bias"]
    assert abs(float(b.std()) - 0.05) < 0.02 and sd["lm_head.weight"] is sd["embed_tokens.weight"]
    w = synthetic.normal("k", 0, (200000,))
    assert abs(w.mean()) < 0.01 and abs(w.std() - 1) < 0.01


def test_state_dict_keys_are_the_reference_275():
    from rnamsm.model import MSATransformer
    m = MSATransformer(num_layers=10)
    keys = set(m.state_dict().keys())
    spec = {k for k, _, _ in synthetic.state_dict_spec()}
    assert len(keys) == 275 and keys == spec
    for k, shape, _ in synthetic.state_dict_spec():
        assert tuple(m.state_dict()[k].shape) == shape, k
    sd = {k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(seed=0, embed_dim=128, num_layers=1, num_heads=2).items()}
    small = MSATransformer(embed_dim=128, num_attention_heads=2, num_layers=1)
    small.load_state_dict(sd, strict=True)
    bad = dict(sd); bad.pop("layers.0.feed_forward_layer.layer.fc1.bias")
    with pytest.raises(RuntimeError):
        small.load_state_dict(bad, strict=True)


# ------------------------------------------------------------------ config / CLI helpers
def test_config_defaults_and_overrides():
    c = Config()
    assert (c.data.device, c.data.MSA_path, c.data.MSA_list, c.data.max_seqlen, c.data.max_tokens,
            c.data.max_seqs_per_msa, c.data.sample_method, c.data.architecture) == \
        ("cuda", "results", "rna_id.txt", 1024, 16384, 512, "hhfilter", "rna language")
    assert (c.model.embed_dim, c.model.num_attention_heads, c.model.num_layers, c.model.gemm_dtype) == (768, 12, 10, "f32")
    c = parse_overrides(["data.root_path=/x", "data.MSA_path=r2", "data.model_path=/m.ckpt", "data.MSA_list=ids.txt",
                         "model.num_layers=3", "data.max_seqs_per_msa=64", "model.embed_positions_msa=false"])
    assert c.data.root_path == "/x" and c.data.MSA_path == "r2" and c.model.num_layers == 3
    assert c.data.max_seqs_per_msa == 64 and c.model.embed_positions_msa is False
    assert Config().data.batch_small_msas_16bit is False              # ... for the exact path; the 16-bit modes batch on request only
    assert parse_overrides(["data.batch_small_msas_16bit=true"]).data.batch_small_msas_16bit is True
    assert Config().data.batch_small_msas is True                     # the extra CLI switch: on by default (round 3) ...
    assert parse_overrides(["data.batch_small_msas=false"]).data.batch_small_msas is False
    with pytest.raises(ValueError):
        parse_overrides(["data.batch_small_msas=maybe"])
    for bad in (["data.nope=1"], ["nogroup.x=1"], ["data.max_seqlen"], ["data.max_seqlen=abc"]):
        with pytest.raises((KeyError, ValueError)):
            parse_overrides(bad)


def test_cli_file_discovery_and_crop(tmp_path):
    from rnamsm.inference import crop_tokens, find_msa_files
    (tmp_path / "A.a2m_msa2").write_text(">a\nACGU\n")
    (tmp_path / "B.extra.a2m_msa2").write_text(">b\nACGU\n")
    found = find_msa_files(tmp_path, ["A", "B"])
    assert sorted(found) == ["A", "B"]
    with pytest.raises(FileNotFoundError):
        find_msa_files(tmp_path, ["A", "C"])
    with pytest.raises(FileNotFoundError):
        find_msa_files(tmp_path / "missing", ["A"])
    with pytest.raises(ValueError):
        find_msa_files(tmp_path, [])
    toks = synthetic.make_tokens(3, 1500, 0)
    out = crop_tokens(toks, 1024, np.random.RandomState(0))
    assert out.shape == (3, 1024) and (out[:, 0] == 0).all()
    assert crop_tokens(toks[:, :1024], 1024, np.random.RandomState(0)).shape == (3, 1024)
    edge = crop_tokens(synthetic.make_tokens(2, 1025, 0), 1024, np.random.RandomState(0))   # reference raises here
    assert edge.shape == (2, 1024)


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_symbol_declared_in_the_header(lib):
    from rnamsm import _lib
    header = open(os.path.join(ROOT, "include", "rnamsm.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rnamsm_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rnamsm_version() == 600         # ABI 6.0 (round 6): same entry points; ten knob names removed, two refusals added (include/rnamsm.h)


def test_no_kernel_of_the_shipped_library_spills(lib):
    """Every gfx950 kernel in librnamsm_hip.so, from the code objects' own metadata (tools/code_objects.py): no VGPR spill and
    no scratch at all -- a default-path kernel that spills re-reads its own registers from memory inside the loop the roofline
    fraction is quoted on.  (SGPR spills go to VGPR lanes, not memory, and are only reported by the tool.)"""
    import importlib.util
    from rnamsm import _lib
    spec = importlib.util.spec_from_file_location("code_objects", os.path.join(ROOT, "tools", "code_objects.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ks = mod.kernels_of(_lib.LIB_PATH)
    assert len(ks) > 100, len(ks)
    bad = [(k["name"], k["vgpr_spills"], k["scratch_bytes"]) for k in ks if k["vgpr_spills"] or k["scratch_bytes"]]
    assert not bad, bad


def test_ctypes_signatures_match_the_header_prototypes():
    """Every prototype of include/rnamsm.h against the ctypes table of rnamsm/_lib.py: same number of parameters, and per
    parameter the same class (pointer / 64-bit integer / int / float / size_t) -- a drifted binding would otherwise pass
    garbage through the C ABI silently (ctypes does not check)."""
    import ctypes
    from rnamsm import _lib
    header = open(os.path.join(ROOT, "include", "rnamsm.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = dict(re.findall(r"\b(rnamsm_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S))
    assert set(protos) == set(_lib._SIGNATURES)

    def klass(decl: str) -> str:
        decl = " ".join(decl.split())
        if "*" in decl or decl.endswith("]"):
            return "ptr"
        base = decl.rsplit(" ", 1)[0] if " " in decl else decl
        return {"int64_t": "i64", "long long": "i64", "size_t": "size", "int": "int", "float": "float",
                "double": "double"}[base.replace("const ", "")]

    ctype_class = {ctypes.c_void_p: "ptr", ctypes.c_char_p: "ptr", ctypes.c_int64: "i64", ctypes.c_longlong: "i64",
                   ctypes.c_size_t: "size", ctypes.c_int: "int", ctypes.c_float: "float", ctypes.c_double: "double"}
    for name, params in protos.items():
        params = params.strip()
        want = [] if params in ("", "void") else [klass(x) for x in params.split(",")]
        got = []
        for t in _lib._SIGNATURES[name][1]:
            got.append("ptr" if hasattr(t, "_type_") and not isinstance(t._type_, str) else ctype_class[t])
        assert got == want, (name, got, want)


def test_library_validates_arguments_without_a_gpu(lib):
    from rnamsm import _lib
    # deterministic function of the shape; an fp32 slab covers at most 32 rows, which the kernel accumulates as four chains
    # of 8 rows = 512 terms (row_split.h)
    assert lib.rnamsm_row_logits_nsplit(256, 512, 12) == 8
    assert lib.rnamsm_row_logits_workspace_bytes(256, 512, 12) == 8 * 12 * 512 * 512 * 4
    for R in (1, 7, 8, 9, 100, 512, 1024):
        n = lib.rnamsm_row_logits_nsplit(R, 36, 12)
        assert n >= (R + 31) // 32 and -(-R // n) <= 32, (R, n)
    dims = _lib.ModelDims(10, 768, 12, 3072, 12, 1026, 1, 1e-5)
    import ctypes
    need = lib.rnamsm_forward_workspace_bytes(ctypes.byref(dims), 256, 512, 0, 0)
    T = 256 * 512
    assert lib.rnamsm_row_logits16_nsplit(256, 512, 12, 3) == 16    # 256x256 tiles, one block per CU: 12 * 16 * 4 = 3 rounds
    assert lib.rnamsm_row_logits16_nsplit(256, 512, 12, 1) == 5     # plain bf16, C % 8 == 0: persistent 256x256 kernel, 240 tiles, few slabs
    assert lib.rnamsm_row_logits16_nsplit(1024, 1024, 12, 1) == 4   # 12 heads x 16 tiles x 4 slabs = 3 rounds of 256 blocks
    assert lib.rnamsm_row_logits16_nsplit(256, 100, 12, 3) == 37     # small C: 128x128 tiles, split by block count only
    assert lib.rnamsm_row_logits16_workspace_bytes(256, 512, 12) == 16 * 12 * 512 * 512 * 4
    assert need >= T * 768 * 4 * 6 + 16 * 12 * 512 * 512 * 4 + T and need < T * 768 * 4 * 7.2
    rc = lib.rnamsm_gemm_bias_act_res(None, 0, None, None, None, 0, None, 0, 4, 128, 32, 0, 1.0, 0, None, 0, None)
    assert rc == -1 and b"null pointer" in lib.rnamsm_last_error()
    rc = lib.rnamsm_gemm_bias_act_res(16, 32, 16, None, None, 0, 16, 100, 4, 100, 32, 0, 1.0, 0, None, 0, None)
    assert rc == -1 and b"N % 128" in lib.rnamsm_last_error()
    rc = lib.rnamsm_gemm_bias_act_res(16, 32, 16, None, None, 0, 16, 128, 4, 128, 32, 0, 1.0, 0, None, 1, None)
    assert rc == -2                                                  # RNAMSM_BF16 reserved
    # the hi/lo-bf16 mode ("bf16x3", dtype 2 / split 3 with fmt 0) was removed in round 5: UNSUPPORTED, not a silent other mode
    rc = lib.rnamsm_gemm_bf16(16, 64, 16, 16, None, None, 0, 16, 128, 4, 128, 64, 0, 1.0, 0, 3, 0, None, None, None, None, None)
    assert rc == -2 and b"bf16x3" in lib.rnamsm_last_error()
    assert "bf16x3" not in _lib.DTYPES
    rc = lib.rnamsm_col_attn_fused(16, 16, 16, 64, 16, 64, 4, 4, 1, 32, None, None, None, 0, 0, None)
    assert rc == -1 and b"head_dim" in lib.rnamsm_last_error()
    rc = lib.rnamsm_embed_ln(16, 16, 16, 16, 16, 16, 16, 1025, 4, 768, 12, 1026, 1, 1e-5, None, None)
    assert rc == -1 and b"maximum MSA depth of 1024" in lib.rnamsm_last_error()


def test_no_cpu_fallback_exists():
    from rnamsm import _lib, ops
    from rnamsm.model import MSATransformer
    with pytest.raises(_lib.RnamsmError):
        ops.linear(torch.zeros(4, 32), torch.zeros(128, 32))
    with pytest.raises(_lib.RnamsmError):
        ops.layernorm(torch.zeros(4, 128), torch.ones(128), torch.zeros(128))
    m = MSATransformer(embed_dim=128, num_attention_heads=2, num_layers=1).eval()
    with pytest.raises(_lib.RnamsmError):
        m(torch.zeros(1, 2, 5, dtype=torch.int64), need_head_weights=True)
    # nothing under the product package imports the oracle
    pkg = os.path.join(ROOT, "rna-msm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().replace("SURVEY", ""), os.path.join(dirpath, f)
    for f in ("RNA_MSM_Inference.py",):
        assert "oracle" not in open(os.path.join(ROOT, f)).read()
    # ... and neither does any helper under tools/ (diagnostics that need the oracle live in tests/analysis/)
    for f in sorted(os.listdir(os.path.join(ROOT, "tools"))):
        path = os.path.join(ROOT, "tools", f)
        if os.path.isfile(path):
            text = open(path).read()
            assert "import oracle" not in text and "from oracle" not in text and "import truth" not in text, path


def test_mask_and_repeat_matches_reference_semantics():
    """utils/likelihood.py:18-27: copy i of the alignment has <mask> at (row 0, indices[i]) and nothing else changes."""
    import torch
    from rnamsm.likelihood import mask_and_repeat
    toks = torch.arange(3 * 7).view(3, 7)
    out = mask_and_repeat(toks, [2, 5, 6, 2], mask_idx=99)
    assert out.shape == (4, 3, 7)
    for i, c in enumerate([2, 5, 6, 2]):
        want = toks.clone(); want[0, c] = 99
        assert torch.equal(out[i], want)
    assert torch.equal(toks, torch.arange(3 * 7).view(3, 7))          # the input is not modified


def test_argument_validation_layer_under_address_and_ub_sanitizers():
    """SURVEY.md §5 (CPU-side hygiene): a HOST-ONLY build of every entry point (`--offload-host-only`: no device code)
    under AddressSanitizer + UBSan (flags in rna-msm_amd/csrc/asan.mk), driven by tests/abi/abi_driver.c with null, misaligned, out-of-range and extreme
    arguments -- every call is refused by the entry checks or is a pure host function, so no GPU is needed.  The make
    target builds the library and the driver and runs it; a sanitizer report or a failed expectation fails the run."""
    import shutil
    import subprocess
    if shutil.which(os.environ.get("HIPCC", "hipcc")) is None:
        pytest.skip("hipcc not on PATH")
    if not os.path.exists(os.path.join(ROOT, "rna-msm_amd", "csrc", "asan.mk")):
        pytest.skip("asan.mk is not shipped to GPU boxes (sanitizers are a CPU-side tool on this pool)")
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "rna-msm_amd", "csrc"), "-j8", "check-asan"], capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "ABI driver:" in p.stdout and " 0 failed" in p.stdout
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr


def test_bench_launcher_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` started directly spawns its two ranks from a parent that never touches the GPU; on this
    GPU-less container every rank refuses to run (there is no CPU path) and the launcher must return non-zero instead of
    hanging or printing a result line."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a GPU-less host")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert "no CPU path" in p.stderr and not any(ln.startswith("{") for ln in p.stdout.splitlines())


def test_bench_compact_line_fits_the_drivers_parser_whatever_the_run_measured():
    """BENCH_r05.json came back `parsed: null` because bench.py's single line had grown to 20 KB.  The line
    the driver parses is now bench.compact_line(result): <= 4096 bytes, a JSON round trip, every key of the bench contract plus
    `roofline` / `cpu_baseline` with their evidence fields and the digest LAST.  Canned input: a complete round-5 result (20 KB,
    tests/golden/bench_detail_canned.json) and a worst case built from it (8 ranks, long free-text fields, per-rank lists)."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    canned = json.loads(open(os.path.join(ROOT, "tests", "golden", "bench_detail_canned.json")).read().strip().splitlines()[-1])
    assert len(json.dumps(canned)) > 15000                              # the input really is the oversized one
    worst = json.loads(json.dumps(canned))
    worst["n_gpus"] = 8
    worst["config"]["gather"] = "failed: " + "RuntimeError: NCCL error in: some/very/long/path.cpp:1234, unhandled system error " * 20
    worst["config"]["workload"] += " " + "x" * 2000
    worst["config"]["ranks_seen"] = [f"rank {r}: cuda:{r} AMD Instinct MI355X uuid {'ab' * 16} pci 0:{r}:0 (8 visible)" for r in range(8)]
    worst["config"]["distinct_devices"] = 8
    worst["gather_stats"] = {"per_rank_bytes_received": [14.3e9] + [0] * 7, "per_rank_host_wait_s": [0.123456789] * 8,
                             "per_rank_stream_wait_ms": [12.3456789] * 8, "free_text": "y" * 5000}
    worst["compute_only_value"], worst["compute_only_ms_per_step"] = 5.1e6, 25.7
    worst["output_digest"] = {"value": 2 ** 63 + 12345, "items": 512, "what": "z" * 500}
    worst["cpu_baseline"]["sample"] = "s" * 3000
    for res in (canned, worst):
        line = bench.compact_line(res)
        text = json.dumps(line)
        assert len(text.encode()) <= bench.COMPACT_LIMIT == 4096, len(text)
        back = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline", "digest"):
            assert k in back, k
        assert list(back)[-1] == "digest" and back["digest"]["value_k"] == canned["digest"]["value_k"]
        for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches", "flops_per_launch", "traffic",
                  "algorithmic_bytes_per_launch", "traffic_source"):
            assert k in back["roofline"], k
        for k in ("value", "unit", "cores", "kind", "sample", "cpu_model", "full_forward_s"):
            assert k in back["cpu_baseline"], k
        for k in ("workload", "gather", "world_size_initialised", "distinct_devices"):
            assert k in back["config"], k
        assert back["torch_rocm_eager"]["value"] > 0
        assert abs(back["value"] - res["value"]) <= 1e-8 * res["value"] and back["n_gpus"] == res["n_gpus"]
        assert abs(back["roofline"]["frac"] - res["roofline"]["frac"]) < 1e-5
    assert back["config"]["distinct_devices"] == 8 and len(back["gather_stats"]["per_rank_bytes_received"]) == 8
    assert back["output_digest"]["value"] == 2 ** 63 + 12345 and back["compute_only_value"] == 5.1e6


def test_reader_agrees_with_the_oracle_on_random_alignments():
    """Differential test of the product's reader/tokenizer (byte LUT, own FASTA parser) against the oracle's restatement
    (regex clean-up + digitize, the reference's method) on random a2m text: every character class (upper-case residues,
    T, IUPAC codes, lower-case / '.' / '*' insertions, gaps), multi-line records, blank lines, descriptions with spaces --
    token matrices bit-identical, and the same exception type on invalid characters and ragged rows."""
    from hypothesis import given, settings, strategies as st
    from rnamsm.msa import read_fasta_records
    a = RNAAlphabet()
    keep = "ACGUX-TRYKMSWBDHVN"
    drop = "acgu.*nxz"

    @settings(max_examples=150, deadline=None)
    @given(st.integers(1, 6), st.integers(1, 40), st.randoms(use_true_random=False), st.sampled_from(["ok", "bad_char", "ragged"]))
    def run(rows, cols, rnd, kind):
        seqs = []
        for r in range(rows):
            s = "".join(rnd.choice(keep) for _ in range(cols))
            pieces = []
            for ch in s:                                           # sprinkle insertions that the reader must drop
                if rnd.random() < 0.2:
                    pieces.append(rnd.choice(drop))
                pieces.append(ch)
            seqs.append("".join(pieces))
        if kind == "bad_char":
            r = rnd.randrange(rows)
            seqs[r] = seqs[r][:1] + rnd.choice("EFIJLOPQZ@5") + seqs[r][1:]
        if kind == "ragged" and rows > 1:
            seqs[-1] = seqs[-1] + "A"
        text = ""
        for r, s in enumerate(seqs):
            cut = rnd.randrange(1, len(s) + 1)
            text += f">seq{r} some description\\n{s[:cut]}\\n" + (f"{s[cut:]}\\n" if cut < len(s) else "") + ("\\n" if rnd.random() < 0.3 else "")
        try:
            want = TO.encode_msa(text)
            err = None
        except (AssertionError, ValueError) as e:
            want, err = None, type(e)
        recs = read_fasta_records(text, is_text=True)
        if err is None:
            got = a.encode_a2m_records([s for _, s in recs])
            assert got.dtype == np.int64 and np.array_equal(got, want)
        else:
            with pytest.raises(err):
                a.encode_a2m_records([s for _, s in recs])

    run()


def test_cli_grouping_rule_for_small_alignments():
    """data.batch_small_msas: a group's frame stays within 32 k tokens and twice its real tokens, at most 64 members."""
    from rnamsm.inference import FRAME_TOKENS, GROUP_MEMBERS, SMALL_MSA_TOKENS, joins_group
    assert (SMALL_MSA_TOKENS, FRAME_TOKENS, GROUP_MEMBERS) == (3072, 32768, 64)
    assert joins_group([], (3, 20)) and joins_group([(3, 20)], (4, 22))
    assert not joins_group([(2, 10), (2, 10)], (30, 50))              # 3 x 30 x 50 = 4500 > 2 x 1540: too much padding
    assert joins_group([(30, 50)] * 20, (30, 50)) and not joins_group([(30, 50)] * 21, (30, 50))    # 22 x 1500 > 32768
    assert joins_group([(2, 8)] * 63, (2, 8)) and not joins_group([(2, 8)] * 64, (2, 8))
    # the maps scale with members x columns^2, not with the tokens: shallow, long alignments stay (nearly) alone
    from rnamsm.inference import FRAME_MAP_ELEMS
    assert FRAME_MAP_ELEMS == 1 << 20
    assert joins_group([(1, 1024)], (1, 1024)) is False               # 2 x 1024^2 maps: 1 GB of atp for 2 k tokens
    assert joins_group([(3, 256)] * 15, (3, 256)) and not joins_group([(3, 256)] * 16, (3, 256))


def test_pooled_small_alignments_are_grouped_by_shape():
    """plan_groups: every pooled alignment lands in exactly one group, groups obey joins_group's limits, and sorting by shape
    needs far fewer groups than taking the list in order (the population of tools/cli_throughput.py's "tiny" case)."""
    from rnamsm.inference import FRAME_TOKENS, GROUP_MEMBERS, joins_group, plan_groups
    rng = np.random.RandomState(1)
    shapes = [(int(rng.randint(2, 13)), int(rng.randint(40, 81)) + 1) for _ in range(64)]
    groups = plan_groups(shapes)
    assert sorted(j for g in groups for j in g) == list(range(64))
    for g in groups:
        assert g == sorted(g) and 1 <= len(g) <= GROUP_MEMBERS
        frame = len(g) * max(shapes[j][0] for j in g) * max(shapes[j][1] for j in g)
        assert frame <= FRAME_TOKENS and (len(g) == 1 or frame <= 2 * sum(shapes[j][0] * shapes[j][1] for j in g))
        assert len(g) == 1 or len(g) * max(shapes[j][1] for j in g) ** 2 <= (1 << 20)
    in_order, cur = 0, []
    for sh in shapes:                                       # the consecutive rule (still used under gather_to_rank0)
        if cur and not joins_group(cur, sh):
            in_order, cur = in_order + 1, []
        cur.append(sh)
    in_order += 1
    assert len(groups) <= 4 < in_order                       # (list order: 6 groups at the 32 k / 64 limits, 12 at 16 k / 32)
    # a wider population (4-24 rows x 40-120 columns): the cheaper of the two plans; frames stay well inside the 2 x rule
    from rnamsm.inference import frame_tokens
    rng = np.random.RandomState(1)
    wide = [(int(rng.randint(4, 25)), int(rng.randint(40, 121)) + 1) for _ in range(64)]
    gw = plan_groups(wide)
    assert sorted(j for g in gw for j in g) == list(range(64))
    for g in gw:
        frame = len(g) * max(wide[j][0] for j in g) * max(wide[j][1] for j in g)
        assert len(g) <= GROUP_MEMBERS and frame <= FRAME_TOKENS and (len(g) == 1 or frame <= 2 * sum(wide[j][0] * wide[j][1] for j in g))
    assert frame_tokens(wide, gw) <= 1.8 * sum(r * c for r, c in wide)
    assert plan_groups([]) == [] and plan_groups([(3, 20)]) == [[0]]
    assert plan_groups([(40, 35), (2, 12), (2, 12), (2, 12), (40, 35)]) == [[1, 2, 3], [0, 4]]          # alike ones meet although the list separates them



def test_token_packed_groups_are_cut_in_list_order_and_balanced():
    """plan_packed_groups / joins_packed (exact mode, data.pack_small_msas): a token-packed group pads nothing, so its bounds are
    what it really holds -- tokens, members, map elements -- and shapes need not match; groups are consecutive runs of the list,
    as few as the bounds allow, holding about the same number of tokens."""
    from rnamsm.inference import FRAME_MAP_ELEMS, PACKED_MEMBERS, PACKED_TOKENS, joins_packed, plan_packed_groups
    assert (PACKED_TOKENS, PACKED_MEMBERS) == (131072, 256)
    assert joins_packed([], (3, 20)) and joins_packed([(2, 10), (2, 10)], (30, 50))         # unlike shapes share a group: nothing is padded
    assert joins_packed([(30, 50)] * 86, (30, 50)) and not joins_packed([(30, 50)] * 87, (30, 50))      # 88 x 1500 > 131072 tokens
    assert joins_packed([(2, 8)] * 255, (2, 8)) and not joins_packed([(2, 8)] * 256, (2, 8))            # members
    assert joins_packed([(3, 256)] * 15, (3, 256)) and not joins_packed([(3, 256)] * 16, (3, 256))      # sum of C^2
    rng = np.random.RandomState(7)
    for n in (1, 2, 63, 64, 65, 70, 256, 300, 700):
        shapes = [(int(rng.randint(1, 25)), int(rng.randint(20, 140))) for _ in range(n)]
        groups = plan_packed_groups(shapes)
        assert [j for g in groups for j in g] == list(range(n))                                # consecutive runs, every position once
        tok = [sum(shapes[j][0] * shapes[j][1] for j in g) for g in groups]
        for g, t in zip(groups, tok):
            assert len(g) <= PACKED_MEMBERS and t <= PACKED_TOKENS and sum(shapes[j][1] ** 2 for j in g) <= FRAME_MAP_ELEMS
        least = max(-(-n // PACKED_MEMBERS), -(-sum(tok) // PACKED_TOKENS), -(-sum(c * c for _, c in shapes) // FRAME_MAP_ELEMS))
        assert len(groups) <= 1.25 * least + 1                                                 # (next-fit in list order: not optimal packing)
        if len(groups) > 1:
            assert min(tok) >= 0.5 * max(tok), tok                                             # balanced: no nearly empty launch set
    assert plan_packed_groups([]) == [] and plan_packed_groups([(3, 20)]) == [[0]]
    assert plan_packed_groups([(40, 35), (2, 12), (2, 12), (2, 12), (40, 35)]) == [[0, 1, 2, 3, 4]]
    assert plan_packed_groups([(1, 1024)] * 3) == [[0], [1], [2]]                              # 1024^2 map elements each: alone
    # exact mode (round 5): alignments on either side of the folded LayerNorm's threshold are planned separately -- every position
    # once, no group mixes the classes
    mixed = [(40, 150), (2, 12), (140, 35), (3, 9), (64, 64), (63, 64), (5, 133)]
    groups = plan_packed_groups(mixed, 4096)
    assert sorted(j for g in groups for j in g) == list(range(len(mixed)))
    for g in groups:
        assert len({mixed[j][0] * mixed[j][1] >= 4096 for j in g}) == 1
    assert plan_packed_groups(mixed, 4096) == [[0, 2, 4], [1, 3, 5, 6]]
    assert plan_packed_groups(mixed, 0) == plan_packed_groups(mixed) == [list(range(len(mixed)))]
    # round 6: a class that would be ONE group of at least split_tokens tokens is dealt into two (the CLI pipelines its groups)
    many = [(8, 60)] * 40                                                               # 19200 tokens: one group ...
    assert plan_packed_groups(many) == [list(range(40))] == plan_packed_groups(many, 0, split_tokens=20000)
    two = plan_packed_groups(many, 0, split_tokens=19200)                               # ... two from the threshold on, evenly filled
    assert [len(g) for g in two] == [20, 20] and sorted(i for g in two for i in g) == list(range(40))
    assert plan_packed_groups([(8, 60)], 0, split_tokens=1) == [[0]]                    # a lone alignment is never split
