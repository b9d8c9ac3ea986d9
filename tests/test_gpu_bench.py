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


def _bench(extra, env=None, expect_rc=0):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True,
                       timeout=900, env=e, cwd=ROOT)
    assert (p.returncode == 0) == (expect_rc == 0), (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    if expect_rc != 0:
        return None
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                     # ONE JSON line, from rank 0
    return json.loads(lines[0])


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


def test_default_workload_runs_weak_scaling_through_the_same_gatherer():
    two = _bench(["--gpus", "2", "--backend", "gloo", "--one-device", "--num-seqs", "16", "--seq-len", "40", "--steps",
                  "2", "--warmup", "1", "--no-cpu-baseline", "--no-fast-mode", "--digest"])
    assert two["scaling"] == "weak" and two["config"]["msas_per_step"] == 2 and two["output_digest"]["items"] == 4


def test_a_dying_rank_fails_the_run_instead_of_hanging():
    _bench(["--gpus", "2", "--backend", "gloo", "--one-device"] + COMMON, env={"RNAMSM_BENCH_FAIL_RANK": "1"}, expect_rc=1)


def test_the_drivers_torchrun_launch_takes_the_same_path():
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
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device"] + COMMON
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    two = json.loads(lines[0])
    one = _bench(["--gpus", "1"] + COMMON)
    assert two["n_gpus"] == 2 and two["config"]["world_size_initialised"] == 2
    assert two["output_digest"]["value"] == one["output_digest"]["value"]
