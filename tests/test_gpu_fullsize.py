"""Parity at BASELINE's full sizes, against an fp64 truth.

For every BASELINE config the HIP forward is compared with the oracle evaluated in float64 (tests/truth.py), next to the
error the reference's OWN arithmetic makes against the same truth (the oracle in float32, resp. bfloat16):

    exact path / f16x3 :  emb rel-L2 <= 1e-4 (north_star)  and  err(HIP) <= 2 x err(oracle fp32 on the CPU)   [emb, atp L2, atp mean]
    bf16 (configs[4])  :  drift(HIP bf16) <= 1.5 x drift(oracle .bfloat16())

i.e. the kernels are held to the reference's own fp32 (bf16) noise at the size in question instead of to a free
tolerance.  Every number is appended to gpurun_out/r06_fullsize_parity.json (copied to profiles/ when committed).
"""
import json
import os

import numpy as np
import pytest
import torch

import truth
from conftest import GOLDEN, ROOT, golden, rel_l2
from rnamsm import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPORT = {}


@pytest.fixture(scope="module")
def model():
    assert torch.cuda.is_available()
    from rnamsm.model import MSATransformer
    m = MSATransformer(num_layers=10)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in truth.state().items()}, strict=True)
    yield m.eval().to(DEV)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    REPORT["_device"] = torch.cuda.get_device_name(0)
    with open(os.path.join(out_dir, "r06_fullsize_parity.json"), "w") as f:
        json.dump(REPORT, f, indent=1)


def hip_outputs(model, tokens, mode):
    model.gemm_dtype = mode
    try:
        out = model.forward_one(torch.from_numpy(tokens).to(DEV))
        assert int(out["err"].item()) == 0
        emb, atp = out["emb"].clone(), out["atp"].clone()
    finally:
        model.gemm_dtype = "f32"
    return emb, atp


def test_device_evaluation_of_the_oracle_is_the_reference_fp64(model):
    """Ties tests/truth.py to the reference: the oracle evaluated in fp64 on the device reproduces the reference's own
    fp64 run (fixture), and evaluated in fp32 there it sits at the CPU oracle's fp32 noise (not some reduced-precision
    matmul mode), on BASELINE configs[1] (M=64, L=128)."""
    for name in ("m8_c17", "m16_c33"):
        g, g64 = golden(f"forward_{name}.npz"), golden(f"forward_{name}_fp64.npz")
        emb, atp = truth.oracle_outputs(g["tokens"], torch.float64, DEV)
        assert rel_l2(emb.cpu().numpy(), g64["emb"]) < 1e-9
        assert np.abs(atp.cpu().numpy() - g64["atp"]).max() < 1e-9
    toks = synthetic.make_tokens(64, 128, 0)
    t_emb, t_atp = truth.oracle_outputs(toks, torch.float64, DEV)
    e_dev = truth.errors(*truth.oracle_outputs(toks, torch.float32, DEV), t_emb, t_atp)
    e_cpu = truth.errors(*truth.oracle_outputs(toks, torch.float32, "cpu"), t_emb, t_atp)
    REPORT["oracle fp32 on device vs on CPU, M=64 L=128"] = {"device": e_dev, "cpu": e_cpu}
    # torch's device fp32 kernels are ~3x noisier than the CPU's here (and 10-30x at the larger sizes): the same order of
    # magnitude, nowhere near a reduced-precision matmul mode (which would be 100x+) -- which is all this checks; the bar of
    # the parity test below is the CPU evaluation
    for k in ("emb_rel_l2", "atp_rel_l2", "atp_mean_abs"):
        assert 0.1 * e_cpu[k] < e_dev[k] < 20.0 * e_cpu[k], (k, e_dev, e_cpu)


def _tokens_2drb1():
    return golden("tokens_2DRB_1_full.npz")["diversity_max_512"]


CASES = [
    ("configs[0] 2DRB_1 1176->512 rows x 36 (diversity-max)", _tokens_2drb1),
    ("configs[1] M=64 L=128", lambda: synthetic.make_tokens(64, 128, 0)),
    ("configs[2] M=256 L=512", lambda: synthetic.make_tokens(256, 512, 0)),
    ("configs[3] M=128 L=256", lambda: synthetic.make_tokens(128, 256, 3)),
    ("configs[4] M=1024 L=1024", lambda: synthetic.make_tokens(1024, 1024, 0)),
]


CPU_YARDSTICK = {"configs[1]"}     # the size the CPU oracle finishes in 2-3 s; configs[0] / [3] (8-20 s of 32 threads, twice that on a box
# with slow host cores) use committed measurements since the end of round 4, like configs[2] / [4]; the `slow` test re-measures them
# configs[4] (M = L = 1024): the CPU oracle needs 12.6 minutes of 32 host threads there, so its errors against the same fp64
# truth were measured ONCE on the GPU box (tests/analysis/yardstick_m1024.py, same tokens, same weights) and committed.
# configs[2] (M = 256, L = 512): 70 s of the same -- measured live in the suite until round 3; since round 4 (the suite's wall-
# clock cap) the default run uses the committed measurement and the `slow` variant below re-measures it.
COMMITTED_YARDSTICK = {"configs[4]": "yardstick_m1024_l1024.json", "configs[2]": "yardstick_m256_l512.json",
                       "configs[0]": "yardstick_2drb1_512x36.json", "configs[3]": "yardstick_m128_l256.json"}


@pytest.mark.slow
@pytest.mark.parametrize("case", [0, 2, 3])
def test_configs2_against_a_live_cpu_yardstick(model, case):
    """-m "gpu and slow": BASELINE configs[0] / [2] / [3] with the reference's own fp32 error measured live on this host (8-70 s of
    32 threads) instead of read from tests/golden/yardstick_*.json, and the committed figure checked against it."""
    label, make = CASES[case]
    toks = make()
    t_emb, t_atp = truth.oracle_outputs(toks, torch.float64, DEV)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    live = truth.errors(*truth.oracle_outputs(toks, torch.float32, "cpu"), t_emb, t_atp)
    y = json.load(open(os.path.join(GOLDEN, COMMITTED_YARDSTICK[label.split()[0]])))["oracle_cpu_f32"]
    for k in ("emb_rel_l2", "atp_rel_l2", "atp_mean_abs"):
        assert 0.8 * y[k] <= live[k] <= 1.25 * y[k], (k, live, y)       # thread blocking moves fp32 sums a little, not the scale
    for mode in ("f32", "f16x3"):
        e = truth.errors(*hip_outputs(model, toks, mode), t_emb, t_atp)
        for k in ("emb_rel_l2", "atp_rel_l2", "atp_mean_abs"):
            assert e[k] <= 2.0 * live[k] + 1e-9, (mode, k, e, live)


@pytest.mark.parametrize("label,make", CASES, ids=[c[0].split()[0] for c in CASES])
def test_every_baseline_config_against_fp64_truth(model, label, make):
    """Yardsticks, all against the same fp64 truth: `oracle_cpu_f32` = the oracle in fp32 on the host's cores, i.e. the
    reference's own arithmetic (PyTorch CPU) at this size -- the bar the HIP exact path and f16x3 are held to (x2);
    `oracle_dev_f32` / `oracle_dev_bf16` = the same code through torch's device kernels (recorded; the bf16 one is the
    reference's .bfloat16() behaviour and bounds the bf16 mode x1.5; the fp32 one is 30-60x noisier than the CPU's and is
    never the bar).  At M = L = 1024 the CPU yardstick is the committed measurement tests/golden/yardstick_m1024_l1024.json
    (emb rel-L2 8.1e-4: the fixed-order row split of the exact path must stay at or below 2x that -- it measures 4.8e-4)."""
    cfg = label.split()[0]
    toks = make()
    t_emb, t_atp = truth.oracle_outputs(toks, torch.float64, DEV)
    rep = {"shape": list(toks.shape)}
    rep["oracle_dev_f32"] = truth.errors(*truth.oracle_outputs(toks, torch.float32, DEV), t_emb, t_atp)
    rep["oracle_dev_bf16"] = e_ref16 = truth.errors(*truth.oracle_outputs(toks, torch.bfloat16, DEV), t_emb, t_atp)
    if cfg in CPU_YARDSTICK:
        torch.set_num_threads(min(32, torch.get_num_threads()))                  # bench.py's sweep: 32 is the fastest
        rep["oracle_cpu_f32"] = truth.errors(*truth.oracle_outputs(toks, torch.float32, "cpu"), t_emb, t_atp)
    if cfg in COMMITTED_YARDSTICK:
        y = json.load(open(os.path.join(GOLDEN, COMMITTED_YARDSTICK[cfg])))
        assert y["shape"] == list(toks.shape) and y["token_seed"] == (3 if cfg == "configs[3]" else 0)
        rep["oracle_cpu_f32"] = y["oracle_cpu_f32"]
        rep["oracle_cpu_f32_source"] = f"tests/golden/{COMMITTED_YARDSTICK[cfg]} ({y['cpu_seconds']:.0f} s of {y['threads']} threads, {y['cpu_model']})"
    e_ref = rep["oracle_cpu_f32"]            # the reference's own fp32 arithmetic (PyTorch CPU): measured here or committed
    for mode in ("f32", "f16x3", "bf16"):
        rep[f"hip_{mode}"] = truth.errors(*hip_outputs(model, toks, mode), t_emb, t_atp)
    REPORT[label] = rep
    print(json.dumps({label: rep}))
    for mode in ("f32", "f16x3"):
        e = rep[f"hip_{mode}"]
        for k in ("emb_rel_l2", "atp_rel_l2", "atp_mean_abs"):
            assert e[k] <= 2.0 * e_ref[k] + 1e-9, (mode, k, e, e_ref)
        assert e["atp_max_abs"] <= max(4.0 * e_ref["atp_max_abs"], 1e-4), (mode, e, e_ref)
        if cfg in ("configs[1]", "configs[2]", "configs[3]"):
            assert e["emb_rel_l2"] <= 1e-4, (mode, e)            # north_star's absolute bar, quoted at M=256 L=512
        elif cfg == "configs[0]":
            # 512 rows x 36 columns: the reference's own fp32 run is this far from the truth, so the absolute bar is
            # that noise floor where it exceeds 1e-4
            assert e["emb_rel_l2"] <= max(1e-4, 2.0 * e_ref["emb_rel_l2"]), (mode, e, e_ref)
    e = rep["hip_bf16"]
    for k in ("emb_rel_l2", "atp_rel_l2", "atp_mean_abs", "atp_max_abs"):
        assert e[k] <= 1.5 * e_ref16[k], ("bf16", k, e, e_ref16)


def test_configs0_through_the_reader_and_device_subsampling(model, full_2drb1_a2m):
    """configs[0] end to end on the device side: the shipped 1176-row alignment file -> reader -> device greedy
    sub-sampling (512 rows) -> forward, equal to the forward of the reference's own 512 selected rows (bit-identical:
    same tokens, deterministic kernels)."""
    from rnamsm.alphabet import RNAAlphabet
    from rnamsm.msa import load_msa_tokens
    toks = load_msa_tokens(full_2drb1_a2m, RNAAlphabet(), 512, "diversity-max", device=DEV)
    want = _tokens_2drb1()
    assert np.array_equal(toks, want)
    a = hip_outputs(model, toks, "f32")
    b = hip_outputs(model, want, "f32")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert a[0].shape == (35, 768) and a[1].shape == (120, 35, 35)                  # shapes of the shipped 2DRB_1_*.npy


@pytest.mark.parametrize("layers,R,L", [(1, 128, 255), (2, 128, 255), (2, 40, 99), (3, 256, 511)])
def test_bf16_mode_at_shallow_depth_against_the_references_bf16_drift(layers, R, L):
    """at ten layers with these weights the bf16 yardstick itself is emb rel-L2 0.25-0.46 and atp max-abs ~ 1 -- a
    bar that catches a broken kernel, not a degraded one.  At one to three layers the reference's own `.bfloat16()` drift is
    1e-2-scale (asserted: the yardstick is far from saturation), and the HIP bf16 mode -- the whole plane data flow, the
    16x16x32 GEMM kernels from 10 k tokens, the rebuilt attention kernels incl. the prescaled-q column kernel -- is held to
    1.5x of it on every measure.  Truth: the oracle in fp64 on the device, same weights."""
    from oracle import msm_oracle as O
    from rnamsm.model import MSATransformer
    state = synthetic.make_state_dict(seed=0, num_layers=layers)
    m = MSATransformer(num_layers=layers)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    m = m.eval().to(DEV)
    toks = synthetic.make_tokens(R, L, 5)

    def oracle(dtype):
        with torch.no_grad():
            res = O.forward(torch.from_numpy(toks).to(DEV), O.to_torch_params(state, dtype, DEV), num_layers=layers, ffn_token_chunk=32768)
            emb, atp = O.pack_outputs(res)
        return emb.double(), atp.double()
    t_emb, t_atp = oracle(torch.float64)
    e_ref = truth.errors(*oracle(torch.bfloat16), t_emb, t_atp)
    e_hip = truth.errors(*hip_outputs(m, toks, "bf16"), t_emb, t_atp)
    print(json.dumps({"layers": layers, "shape": [R, L + 1], "oracle_bf16": e_ref, "hip_bf16": e_hip}))
    assert 1e-4 < e_ref["emb_rel_l2"] < 0.1 and e_ref["atp_rel_l2"] < 0.2, e_ref              # a yardstick with room above AND below
    for k in ("emb_rel_l2", "atp_rel_l2", "atp_mean_abs", "atp_max_abs"):
        assert e_hip[k] <= 1.5 * e_ref[k], (k, e_hip, e_ref)
    e_f16 = truth.errors(*hip_outputs(m, toks, "f16x3"), t_emb, t_atp)                       # and the fp32-grade mode at fp32 scale
    assert e_f16["emb_rel_l2"] < 2e-5 and e_f16["atp_max_abs"] < 2e-4, e_f16
