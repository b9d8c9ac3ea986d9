"""Multi-GPU job farm: independent MSAs are sharded over ranks, outputs gathered to rank 0.

The path shards by independent units: an MSA's forward never communicates, and the
reference's own scale-out is one process per GPU over a list of ids (utils/distribute.py:9-38).  Here it is one
process per GPU with torch.distributed; the only data-path communication is the output gather
(emb [L,768] + atp [120,L,L] per MSA) to rank 0, done with point-to-point sends so that on MI355X every peer uses
its own direct xGMI link to rank 0 (no ring).  With backend "nccl" this is RCCL; the same code runs on "gloo"
for the CPU tests (tests/test_sharding.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_indices(num_items: int, rank: int, world_size: int) -> List[int]:
    """Round-robin assignment: item i -> rank i % world_size (balanced to within one item, and a sorted id list
    keeps neighbouring sequence lengths on different ranks)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, num_items, world_size))


def owner_of(index: int, world_size: int) -> int:
    return index % world_size


def gather_arrays(local: Dict[int, Sequence[torch.Tensor]], num_items: int, dst: int = 0,
                  group: Optional[dist.ProcessGroup] = None) -> Optional[Dict[int, List[torch.Tensor]]]:
    """Gather per-item tensor tuples to `dst`.

    `local` maps the global item index (as assigned by shard_indices) to that item's tensors -- all items carry the
    same number of tensors, shapes may differ per item (L varies).  Shapes travel first as one small int64 message
    per peer, then the payloads as batched isend/irecv.  Returns {index: [tensors]} on dst, None elsewhere."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return {i: list(ts) for i, ts in local.items()}
    mine = shard_indices(num_items, rank, world)
    assert sorted(local) == mine, f"rank {rank} holds {sorted(local)} but owns {mine}"
    any_t = next(iter(local.values()))[0] if local else None
    device = any_t.device if any_t is not None else torch.device("cpu")
    n_tensors = len(next(iter(local.values()))) if local else 0
    # every rank must agree on tensors-per-item and dtype even when it owns nothing
    meta = torch.tensor([n_tensors], dtype=torch.int64, device=device)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=group)
    n_tensors = int(meta.item())
    MAXD = 4

    def shape_msg(items):
        m = torch.zeros(len(items), n_tensors, MAXD + 1, dtype=torch.int64)
        for a, idx in enumerate(items):
            for b, t in enumerate(local[idx]):
                m[a, b, 0] = t.dim()
                m[a, b, 1:1 + t.dim()] = torch.tensor(t.shape, dtype=torch.int64)
        return m.to(device)

    if rank != dst:
        if not mine:
            return None
        dist.send(shape_msg(mine), dst=dst, group=group)
        ops = [dist.P2POp(dist.isend, t.contiguous(), dst, group) for idx in mine for t in local[idx]]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        return None

    out: Dict[int, List[torch.Tensor]] = {i: list(ts) for i, ts in local.items()}
    dtype = any_t.dtype if any_t is not None else torch.float32
    ops, pending = [], []
    for peer in range(world):
        if peer == dst:
            continue
        theirs = shard_indices(num_items, peer, world)
        if not theirs:
            continue
        m = torch.zeros(len(theirs), n_tensors, MAXD + 1, dtype=torch.int64, device=device)
        dist.recv(m, src=peer, group=group)
        m = m.cpu()
        for a, idx in enumerate(theirs):
            bufs = []
            for b in range(n_tensors):
                nd = int(m[a, b, 0])
                buf = torch.empty(tuple(int(v) for v in m[a, b, 1:1 + nd]), dtype=dtype, device=device)
                ops.append(dist.P2POp(dist.irecv, buf, peer, group))
                bufs.append(buf)
            pending.append((idx, bufs))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for idx, bufs in pending:
        out[idx] = bufs
    return out
