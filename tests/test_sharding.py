"""Multi-GPU host logic on CPU: world_size-2 `gloo` processes exercise the same shard / gather code the RCCL path
uses (rnamsm.sharding).  The per-item payloads stand in for (emb [L,768], atp [120,L,L]) with ragged L."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rnamsm import sharding


def test_shard_indices_partition_the_items():
    for n in (0, 1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            parts = [sharding.shard_indices(n, r, world) for r in range(world)]
            assert sorted(i for p in parts for i in p) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            for r, p in enumerate(parts):
                assert all(sharding.owner_of(i, world) == r for i in p)
    with pytest.raises(ValueError):
        sharding.shard_indices(4, 2, 2)


def _item(idx):
    L = 3 + (idx * 5) % 7                                            # ragged lengths
    g = torch.Generator().manual_seed(idx)
    return [torch.randn(L, 8, generator=g), torch.randn(4, L, L, generator=g)]


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        local = {i: _item(i) for i in sharding.shard_indices(n_items, rank, world)}
        out = sharding.gather_arrays(local, n_items, dst=0)
        if rank == 0:
            ok = sorted(out) == list(range(n_items)) and all(
                all(torch.equal(a, b) for a, b in zip(out[i], _item(i))) for i in range(n_items))
            q.put(bool(ok))
        else:
            q.put(out is None)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 5), (2, 1), (2, 8)])
def test_gather_to_rank0_over_gloo(world, n_items):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(results)


def test_single_process_gather_is_identity():
    local = {i: _item(i) for i in range(3)}
    out = sharding.gather_arrays(local, 3)
    assert sorted(out) == [0, 1, 2] and torch.equal(out[1][1], _item(1)[1])
