"""bench.py's multi-rank flow on the one-GPU box: `python bench.py --gpus 2` started directly must
launch its own ranks, shard the configs[3] batch with rnamsm.sharding.shard_indices, gather every output to rank 0 with
rnamsm.sharding.RoundGatherer and print one rank-0 JSON line -- and the gathered outputs must be the N=1 run's, bit for
bit.  RCCL refuses two ranks on one device, so the N=2 run uses the bench's test hooks (--one-device, --backend gloo:
same control flow, payloads staged through the host); the driver's 8-GPU run is the same code on backend nccl."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
COMMON = ["--workload", "configs3", "--num-msas", "6", "--num-seqs", "16", "--seq-len", "40", "--steps", "1",
          "--warmup", "1", "--digest", "--no-cpu-baseline", "--no-fast-mode"]


COMPACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "digest")


def _parse(stdout, detail_path):
    """bench.py's stdout is ONE JSON line, the last one, <= 4096 bytes, carrying the bench contract's keys (what the driver parses:
    VERDICT r05 item 1); the complete result is the detail file.  Returns the detail with the compact line under "_compact"."""
    out_lines = [ln for ln in stdout.splitlines() if ln.strip()]
    lines = [ln for ln in out_lines if ln.startswith("{")]
    assert len(lines) == 1 and out_lines[-1] == lines[0], stdout          # ONE JSON line, from rank 0, and it is the LAST line
    assert len(lines[0].encode()) <= 4096, len(lines[0])
    compact = json.loads(lines[0])
    assert all(k in compact for k in COMPACT_KEYS), sorted(compact)
    assert all(k in compact["roofline"] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")), compact["roofline"]
    assert all(k in compact["config"] for k in ("workload", "gather", "world_size_initialised", "distinct_devices"))
    assert list(compact)[-1] == "digest"
    detail = json.load(open(detail_path))
    for k in ("metric", "n_gpus", "steps", "warmup", "scaling", "dtype"):
        assert compact[k] == detail[k]
    assert abs(compact["value"] - detail["value"]) <= 1e-6 * detail["value"]
    detail["_compact"] = compact
    return detail


def _bench(extra, env=None, expect_rc=0):
    import tempfile
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    with tempfile.TemporaryDirectory(prefix="rnamsm_bench_test_") as tmp:
        detail = os.path.join(tmp, "detail.json")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--detail-out", detail] + extra, capture_output=True,
                           text=True, timeout=900, env=e, cwd=ROOT)
        assert (p.returncode == 0) == (expect_rc == 0), (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
        if expect_rc != 0:
            return None
        return _parse(p.stdout, detail)


def test_the_default_flow_with_both_baselines_prints_one_compact_last_line():
    """The driver's command shape (`bench.py --gpus 1 --steps K --warmup W`, every optional block ON: per-config block, 16-bit modes,
    cpu_baseline, torch_rocm_eager) at a small alignment shape: stdout's LAST line is the one compact JSON line (<= 4096 bytes,
    _parse checks the keys) and it carries `roofline` and `cpu_baseline`; the detail file holds the blocks the line leaves out."""
    res = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--num-seqs", "64", "--seq-len", "128"])
    c = res["_compact"]
    assert c["steps"] == 2 and c["warmup"] == 1 and c["n_gpus"] == 1 and c["dtype"] == "f32" and c["vs_baseline"] is None
    assert c["roofline"]["bound"] == "mfma" and 0 < c["roofline"]["frac"] < 1 and c["roofline"]["launches"] > 0
    assert c["cpu_baseline"]["value"] > 0 and c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] >= 1
    assert c["torch_rocm_eager"]["value"] > 0 and c["digest"]["cpu_res_per_s"] > 0
    assert abs(c["ms_per_step"] * c["steps"] * 1e-3 * c["value"] - 64 * 128 * 2) < 1e-3 * 64 * 128 * 2
    for k in ("per_config", "per_kernel", "fast_mode", "bf16_mode", "kernel_ms_per_msa"):
        assert k in res and k not in c, k


def test_two_ranks_gather_the_same_outputs_as_one_rank():
    one = _bench(["--gpus", "1"] + COMMON)
    two = _bench(["--gpus", "2", "--backend", "gloo", "--one-device"] + COMMON)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["world_size_initialised"] == 2 and two["config"]["backend"] == "gloo"
    assert "RoundGatherer" in two["config"]["gather"] and two["scaling"] == "strong"
    assert one["output_digest"]["items"] == two["output_digest"]["items"] == 6
    assert one["output_digest"]["value"] == two["output_digest"]["value"]          # bit-identical gathered outputs
    for r in (one, two):
        assert r["outputs_finite"] and r["value"] > 0 and r["roofline"]["achieved"] > 0
        assert r["metric"].startswith("MSA-residues/sec") and r["config"]["msas_per_step"] == 6
    # what a first real multi-GPU run needs to be diagnosable: the compute-only curve next to the value,
    # bytes into every rank and the gather's exposed (non-overlapped) time
    assert one["compute_only_value"] is None and one["gather_stats"] is None
    gs = two["gather_stats"]
    assert two["compute_only_value"] > 0 and two["compute_only_ms_per_step"] > 0
    assert len(gs["per_rank_bytes_received"]) == 2 and gs["per_rank_bytes_received"][0] > 0 and gs["per_rank_bytes_received"][1] == 0
    assert len(gs["per_rank_host_wait_s"]) == 2 and len(gs["per_rank_stream_wait_ms"]) == 2


def test_eight_ranks_on_one_device_gather_the_same_outputs_as_one_rank():
    """The driver's largest launch, `--gpus 8`, with the test hooks (all ranks on device 0, gloo), in the DEFAULT suite since round 5
    (VERDICT r04 item 7b): the configs3 strong-scaling workload trimmed to 16 MSAs over 8 ranks -- two full rounds -- gathered
    bit-identical to the N = 1 run; per-rank gather statistics present; every rank's device identity on the line (here: eight
    ranks, ONE distinct device -- the field a real 8-GPU run must show 8 in)."""
    common = ["--workload", "configs3", "--num-msas", "16", "--num-seqs", "8", "--seq-len", "24", "--steps", "1", "--warmup", "1",
              "--digest", "--no-cpu-baseline", "--no-fast-mode"]
    one = _bench(["--gpus", "1"] + common)
    eight = _bench(["--gpus", "8", "--backend", "gloo", "--one-device"] + common)
    assert eight["n_gpus"] == 8 and eight["config"]["world_size_initialised"] == 8
    assert one["output_digest"]["items"] == eight["output_digest"]["items"] == 16
    assert one["output_digest"]["value"] == eight["output_digest"]["value"]
    assert len(eight["config"]["ranks_seen"]) == 8 and eight["config"]["distinct_devices"] == 1
    assert all(f"rank {r}:" in s for r, s in enumerate(eight["config"]["ranks_seen"]))
    assert one["config"]["distinct_devices"] == 1 and len(one["config"]["ranks_seen"]) == 1 and "uuid" in one["config"]["ranks_seen"][0]
    assert eight["digest"]["distinct_devices"] == 1 and len(json.dumps(eight["digest"])) <= 1200
    assert eight["_compact"]["config"]["distinct_devices"] == 1 and eight["_compact"]["output_digest"]["value"] == one["output_digest"]["value"]
    gs = eight["gather_stats"]
    assert len(gs["per_rank_bytes_received"]) == 8 and gs["per_rank_bytes_received"][0] > 0 and sum(gs["per_rank_bytes_received"][1:]) == 0
    assert eight["compute_only_value"] > 0


def test_default_workload_runs_weak_scaling_through_the_same_gatherer():
    two = _bench(["--gpus", "2", "--backend", "gloo", "--one-device", "--num-seqs", "16", "--seq-len", "40", "--steps",
                  "2", "--warmup", "1", "--no-cpu-baseline", "--no-fast-mode", "--digest"])
    assert two["scaling"] == "weak" and two["config"]["msas_per_step"] == 2 and two["output_digest"]["items"] == 4


def test_a_failing_gather_probe_moves_the_gather_to_gloo_and_still_delivers_every_output():
    """a fabric that refuses the point-to-point gather must cost neither the scaling number NOR the gathered
    outputs.  The probe before the timed region is made to fail on every rank (test hook; a refusal at communicator creation looks
    like this): the ranks agree over the gloo control group, move the gather onto it (host-staged) in the same processes, and the
    ONE line is a GATHERED line -- `config.gather` starts with "gloo fallback after: <text>", every output delivered, bit-identical
    to the N = 1 run's."""
    one = _bench(["--gpus", "1"] + COMMON)
    two = _bench(["--gpus", "2", "--backend", "gloo", "--one-device"] + COMMON, env={"RNAMSM_BENCH_FAIL_GATHER_PROBE": "all"})
    assert two["n_gpus"] == 2 and two["value"] > 0 and two["outputs_finite"]
    assert two["config"]["gather"].startswith("gloo fallback after: ") and "injected gather-probe failure" in two["config"]["gather"]
    assert "RoundGatherer" in two["config"]["gather"] and two["digest"]["gather"].startswith("gloo fallback after: ")
    assert two["output_digest"]["items"] == one["output_digest"]["items"] == 6
    assert two["output_digest"]["value"] == one["output_digest"]["value"]
    assert two["gather_stats"]["per_rank_bytes_received"][0] > 0


def test_one_failing_rank_is_enough_to_move_every_rank():
    two = _bench(["--gpus", "2", "--backend", "gloo", "--one-device"] + COMMON, env={"RNAMSM_BENCH_FAIL_GATHER_PROBE": "1"})
    assert two["config"]["gather"].startswith("gloo fallback after: rank 1: ") and two["output_digest"]["items"] == 6


def test_rccl_refusing_two_ranks_on_one_device_is_a_real_refusal_the_fallback_survives():
    """Not injected: backend nccl with both ranks on device 0.  RCCL refuses ("Duplicate GPU detected") -- or, should it accept,
    the primary transport simply works; either way the line is gathered and the outputs are the N = 1 run's bits."""
    one = _bench(["--gpus", "1"] + COMMON)
    two = _bench(["--gpus", "2", "--backend", "nccl", "--one-device"] + COMMON, env={"RNAMSM_BENCH_PG_TIMEOUT_S": "120"})
    assert two["n_gpus"] == 2 and "RoundGatherer" in two["config"]["gather"]
    assert two["config"]["gather"].startswith(("gloo fallback after: ", "rnamsm.sharding.RoundGatherer"))
    assert two["output_digest"]["items"] == 6 and two["output_digest"]["value"] == one["output_digest"]["value"]


def test_when_no_transport_works_the_line_is_the_sharded_compute_alone_and_says_so():
    """Both transports fail (test hook "both"): ONE line, exit code 0, `config.gather` = "failed: <texts>", nothing gathered."""
    two = _bench(["--gpus", "2", "--backend", "gloo", "--one-device"] + COMMON, env={"RNAMSM_BENCH_FAIL_GATHER_PROBE": "both"})
    assert two["n_gpus"] == 2 and two["value"] > 0 and two["outputs_finite"]
    assert two["config"]["gather"].startswith("failed: ") and "injected gather-probe failure" in two["config"]["gather"]
    assert two["digest"]["gather"].startswith("failed: ")
    assert two["output_digest"]["items"] == 0                    # nothing was gathered: the line is the compute alone


def test_a_dying_rank_fails_the_run_instead_of_hanging():
    _bench(["--gpus", "2", "--backend", "gloo", "--one-device"] + COMMON, env={"RNAMSM_BENCH_FAIL_RANK": "1"}, expect_rc=1)


def test_the_drivers_torchrun_launch_takes_the_same_path(tmp_path):
    """The driver starts N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`: bench.py then finds RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment
    and must NOT launch ranks of its own.  Same workload, same digest as the self-launched run; one JSON line from rank 0."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device",
           "--detail-out", str(tmp_path / "detail.json")] + COMMON
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    two = _parse(p.stdout, str(tmp_path / "detail.json"))
    one = _bench(["--gpus", "1"] + COMMON)
    assert two["n_gpus"] == 2 and two["config"]["world_size_initialised"] == 2
    assert two["output_digest"]["value"] == one["output_digest"]["value"]


def test_round_gatherer_on_the_rccl_backend_with_one_rank(tmp_path):
    """The `nccl` (RCCL) configuration of RoundGatherer on the one GPU there is: a world of ONE rank initialises the backend
    the 8-GPU run uses, so default_wire_device() and the device placement of the nccl branch are executed (payloads stay on
    the HIP device, nothing is staged through the host); with one rank no transfer is posted -- the point-to-point branch
    itself still needs a second GPU (SCALE / MULTICHIP runs of the driver)."""
    code = r'''
import os, sys, json
sys.path[:0] = [os.path.join(@ROOT@, "rna-msm_amd"), @ROOT@]
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=@PORT@, HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from rnamsm import sharding
wire = sharding.default_wire_device()
got = {}
g = sharding.RoundGatherer(3, on_item=lambda i, ts: got.__setitem__(i, ts), tensors_per_item=2)
items = [(torch.full((4, 8), float(i), device="cuda:0").t(), torch.arange(2 * i + 2, device="cuda:0", dtype=torch.float32)) for i in range(3)]
for i, ts in enumerate(items):
    g.submit(i, ts)
g.finish()
t = torch.ones(4, device="cuda:0")
dist.all_reduce(t)                                   # the backend really is up: one RCCL collective on the device
ok = (wire.type == "cuda" and g.wire.type == "cuda" and sorted(got) == [0, 1, 2]
      and all(a.is_cuda and torch.equal(a, b) for i in range(3) for a, b in zip(got[i], items[i])) and float(t.sum()) == 4.0)
st = g.stats()
print(json.dumps({"ok": bool(ok), "backend": dist.get_backend(), "stats": st}))
dist.destroy_process_group()
'''
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    code = code.replace("@ROOT@", repr(ROOT)).replace("@PORT@", repr(str(port)))
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["ok"] and res["backend"] == "nccl" and res["stats"]["bytes_received"] == 0


def test_round_gatherer_on_the_rccl_backend_with_two_ranks_on_one_device_or_records_why_not(tmp_path):
    """the point-to-point branch of RoundGatherer (header isend, event-gated batch_isend_irecv, blocking
    recv of the headers on rank 0; sharding.py `_post_round` / `_retire`) has never executed with a peer on backend `nccl`,
    because the boxes this suite sees have ONE GPU and RCCL refuses two ranks on one device.  This test TRIES exactly that --
    two fresh child processes of a GPU-free parent (nothing re-execs after touching the GPU), both on device 0, a bounded wait
    -- and either verifies the gathered tensors bit for bit (the first execution of that branch) or SKIPS with RCCL's own error
    text, so the reason is on record in the run's log instead of in prose."""
    code = r'''
import os, sys, json, traceback
rank = int(sys.argv[1])
sys.path[:0] = [os.path.join(@ROOT@, "rna-msm_amd"), @ROOT@]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=@PORT@, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0")
import datetime
import torch, torch.distributed as dist
try:
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=60))
    t = torch.full((4,), float(rank + 1), device="cuda:0")
    dist.all_reduce(t)                                   # communicator creation happens here: two ranks, one device
    torch.cuda.synchronize()
    assert float(t[0]) == 3.0
except Exception as e:                                    # noqa: BLE001
    print(json.dumps({"refused": f"{type(e).__name__}: {str(e)[:600]}"}), flush=True)
    sys.exit(77)
from rnamsm import sharding
got = {}
n = 5                                                    # rounds 0, 1 full; round 2 with an item-less rank 1
g = sharding.RoundGatherer(n, on_item=lambda i, ts: got.__setitem__(i, [x.clone() for x in ts]), tensors_per_item=2, dst=0,
                           device=torch.device("cuda", 0))
def item(i):
    return (torch.full((3 + i, 8), float(i), device="cuda:0").t(), torch.arange(4 * i + 2, device="cuda:0", dtype=torch.float32) * (i + 1))
for i in sharding.shard_indices(n, rank, 2):
    g.submit(i, item(i))
g.finish()
torch.cuda.synchronize()
ok = True
if rank == 0:
    ok = sorted(got) == list(range(n)) and all(torch.equal(a, b) for i in range(n) for a, b in zip(got[i], item(i)))
print(json.dumps({"ok": bool(ok), "rank": rank, "stats": g.stats()}), flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    code = code.replace("@ROOT@", repr(ROOT)).replace("@PORT@", repr(str(port)))
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e, cwd=ROOT)
             for r in range(2)]
    outs = []
    timed_out = False
    for p in procs:
        try:
            outs.append(p.communicate(timeout=180))
        except subprocess.TimeoutExpired:
            timed_out = True
            p.kill()                                       # the exact PIDs this test started
            outs.append(p.communicate())
    texts = [ln for o, _ in outs for ln in o.splitlines() if ln.startswith("{")]
    refused = [json.loads(t)["refused"] for t in texts if "refused" in t]
    if refused or timed_out or any(p.returncode == 77 for p in procs):
        why = refused[0] if refused else ("no answer within 180 s (the ranks were ended)" if timed_out else (outs[0][1] or outs[1][1])[-600:])
        pytest.skip(f"RCCL does not run two ranks on one device here, the nccl point-to-point branch stays unexecuted: {why}")
    assert all(p.returncode == 0 for p in procs), [(p.returncode, o[-1500:], er[-3000:]) for p, (o, er) in zip(procs, outs)]
    res = [json.loads(t) for t in texts]
    assert all(r["ok"] for r in res) and any(r["rank"] == 0 and r["stats"]["bytes_received"] > 0 for r in res)
