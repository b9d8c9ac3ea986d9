"""Multi-GPU job farm: independent MSAs are sharded over ranks, outputs gathered to rank 0.

The path shards by independent units: an MSA's forward never communicates, and the
reference's own scale-out is one process per GPU over a list of ids (utils/distribute.py:9-38).  Here it is one
process per GPU with torch.distributed; the only data-path communication is the output gather
(emb [L,768] + atp [120,L,L] per MSA) to rank 0, done with point-to-point sends so that on MI355X every peer uses
its own direct xGMI link to rank 0 (no ring).  With backend "nccl" this is RCCL; the same code runs on "gloo"
(which moves host memory: payloads are staged through the host) for the CPU tests (tests/test_sharding.py).

`RoundGatherer` is the one gather implementation (the CLI, bench.py and `gather_arrays` all use it).  It works in
ROUNDS of `world` items -- round k holds items k*world .. k*world+world-1, one per rank -- so that at most one round of
outputs is resident on the destination at a time (atp is 126 MB per L=512 MSA: gathering a whole id list first would
exhaust rank 0's HBM), and it issues round k's transfers on a side stream behind an event, so they overlap the forward
of round k+1.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

_DTYPES = (torch.float32, torch.float64, torch.int64, torch.int32, torch.uint8, torch.bfloat16, torch.float16,
           torch.int16)
_MAXD = 4


def shard_indices(num_items: int, rank: int, world_size: int) -> List[int]:
    """Round-robin assignment: item i -> rank i % world_size (balanced to within one item, and a sorted id list
    keeps neighbouring sequence lengths on different ranks)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, num_items, world_size))


def owner_of(index: int, world_size: int) -> int:
    return index % world_size


def default_wire_device(group: Optional[dist.ProcessGroup] = None) -> torch.device:
    """Memory the backend moves: the current HIP device under RCCL ("nccl"), host memory otherwise.  Never derived
    from the payload, so a rank that owns no item still enters the collectives with the right kind of tensor."""
    if dist.is_initialized() and dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def negotiate_gather_transport(device: torch.device, want_backend: str = "nccl", inject_failure: Optional[str] = None,
                               timeout=None):
    """Decide ONCE, before any output travels, which transport the gather uses -- so that a fabric (or an RCCL build) that refuses
    point-to-point traffic costs the overlap with compute, not the gathered outputs.  Collective: every
    rank calls it.  The DEFAULT process group must be `gloo`: it is the control plane (agreement, barriers, timing reductions
    never depend on the transport under test) and the second transport.

    1. want_backend "nccl": an RCCL group is created in the same processes (`dist.new_group(backend="nccl")`, no restart, nothing
       re-exec'ed) and the gather is probed on it (this also builds the point-to-point communicators); "gloo": the probe runs
       on the default group;
    2. the ranks agree on the outcome over the default gloo group;
    3. if any rank failed, the probe is repeated on the default gloo group (payloads staged through host memory), agreed again.

    Returns (group, label, failure): group = the RCCL group | None (the default gloo group) | False (no transport: compute only);
    label = "primary" | "gloo fallback after: <first failure text>" | "failed: <text>"; failure = this rank's exception text or None.
    `inject_failure` (test hook) makes the primary probe raise on this rank."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if dist.get_backend() != "gloo":
        raise RuntimeError("negotiate_gather_transport: the default process group must be gloo (control plane + second transport)")

    def attempt(step) -> Optional[str]:
        try:
            step()
            return None
        except Exception as e:                                        # noqa: BLE001 -- whatever the backend raises is the answer
            return f"{type(e).__name__}: {e}"

    def agree(mine: Optional[str]):
        texts = [None] * world
        dist.all_gather_object(texts, mine)
        bad = [f"rank {r}: {str(t)[:300]}" for r, t in enumerate(texts) if t]
        return bad[0] if bad else None

    def gather_once(group):
        g = RoundGatherer(world, tensors_per_item=2, dst=0, group=group, device=device)
        g.submit(rank, (torch.full((3, 5), float(rank), device=device), torch.zeros(2, 4, 4, device=device)))
        g.finish()
        if device.type == "cuda":
            torch.cuda.synchronize(device)

    # Two agreements per transport: (a) the group exists on every rank -- a rank that cannot even create it must say so BEFORE
    # anybody blocks in a receive from it; (b) the point-to-point gather itself ran (a failure that shows only here is symmetric
    # in practice; an asymmetric one costs the peers the group's timeout, then they land in the same fallback).
    primary = [None]

    def create():
        if want_backend == "nccl":
            # blocking wait (read when the group is constructed): a transfer that does not complete within the group's timeout
            # RAISES in the waiting rank -- which lands in the fallback below -- instead of the watchdog tearing the process down;
            # the gather's waits are host-side anyway (RoundGatherer._retire runs after the next forward has been enqueued)
            import os
            os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
            primary[0] = dist.new_group(backend="nccl", timeout=timeout) if timeout is not None else dist.new_group(backend="nccl")
        if inject_failure:
            raise RuntimeError(inject_failure)

    mine = attempt(create)
    first = agree(mine)
    if first is None:
        mine = attempt(lambda: gather_once(primary[0]))
        first = agree(mine)
        if first is None:
            return primary[0], "primary", None
    second = agree(attempt(lambda: gather_once(None)))
    if second is None:
        return None, f"gloo fallback after: {first}", mine
    return False, f"failed: {first}; gloo fallback: {second}", mine


class RoundGatherer:
    """Streams per-item tensor tuples to `dst`, one round of `world` items at a time.

    Every rank calls `submit(index, tensors)` for each of its items in ascending order (its k-th call is item
    k*world + rank) and `finish()` once at the end.  On `dst`, `on_item(index, tensors)` is called for every item of
    the job -- its own and the received ones -- in ascending index order, one round behind the submissions (round k is
    delivered when round k+1 is submitted, or by finish()), with tensors on `device`; nothing is retained afterwards.
    Items may differ in shape (L varies per MSA): a small int64 header travels ahead of each payload.
    `tensors_per_item` is static so that ranks owning nothing agree without a collective."""

    def __init__(self, num_items: int, on_item: Optional[Callable[[int, List[torch.Tensor]], None]] = None,
                 tensors_per_item: int = 2, dst: int = 0, group: Optional[dist.ProcessGroup] = None,
                 device: Optional[torch.device] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.dst = dst
        self.num_items = num_items
        self.n_tensors = tensors_per_item
        self.on_item = on_item
        self.wire = default_wire_device(group)
        self.device = torch.device(device) if device is not None else self.wire
        self.num_rounds = (num_items + self.world - 1) // self.world
        self._round = 0
        self._inflight = None            # (round, works, keepalive, [(index, tensors)])
        self._side = torch.cuda.Stream(self.wire) if (self.world > 1 and self.wire.type == "cuda") else None
        self.bytes_received = 0
        # how much of the gather was NOT hidden behind compute (diagnostics for the first real multi-GPU run, bench.py):
        # host seconds this rank spent blocked in _retire, and -- device wire -- event pairs around the compute stream's wait
        # for the transfer stream (read by stats(), which synchronises)
        self.host_wait_s = 0.0
        self._wait_events = []
        self._wait_ms_folded = 0.0         # elapsed time of event pairs already folded (the list stays short on long id lists)

    # ------------------------------------------------------------------ helpers
    def _has_item(self, rnd: int, rank: int) -> bool:
        return rnd * self.world + rank < self.num_items

    def _header(self, tensors: Sequence[torch.Tensor]) -> torch.Tensor:
        h = torch.zeros(self.n_tensors, _MAXD + 2, dtype=torch.int64)
        for b, t in enumerate(tensors):
            if t.dim() > _MAXD:
                raise ValueError(f"gather supports tensors of up to {_MAXD} dimensions")
            h[b, 0] = _DTYPES.index(t.dtype)
            h[b, 1] = t.dim()
            for a, n in enumerate(t.shape):
                h[b, 2 + a] = n
        return h

    def _retire(self) -> None:
        """Complete the round in flight: wait for its transfers, hand its items to on_item (dst), drop the buffers."""
        if self._inflight is None:
            return
        _, works, keep, items = self._inflight
        self._inflight = None
        t0 = time.perf_counter()
        cur = torch.cuda.current_stream(self.wire) if self._side is not None else None
        if cur is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
        for w in works:
            w.wait()
        if cur is not None:
            cur.wait_stream(self._side)
            e1.record(cur)
            self._wait_events.append((e0, e1))
            if len(self._wait_events) > 64:                           # fold the pairs that have completed; never grows past this
                self._fold_wait_events(keep_pending=True)
        self.host_wait_s += time.perf_counter() - t0
        if self.rank == self.dst and self.on_item is not None:
            for index, tensors in sorted(items, key=lambda it: it[0]):
                self.on_item(index, [t if t.device == self.device else t.to(self.device) for t in tensors])
        del keep, items

    def _post_round(self, rnd: int, own: Optional[Sequence[torch.Tensor]]) -> None:
        index = rnd * self.world + self.rank
        works, keep, items = [], [], []
        if self.world == 1:
            self._inflight = (rnd, works, keep, [(index, list(own))])
            return
        ev = payload = None
        if self.rank != self.dst and own is not None:
            # made contiguous / moved to the wire HERE, on the producer's stream (behind the kernels that wrote the tensors),
            # never on the side stream ahead of the event: a strided payload would otherwise be copied while its forward may
            # still be running
            payload = [t.contiguous().to(self.wire) for t in own]
            if self._side is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.wire))  # the payload is complete once this event has fired

        def comm():
            if self.rank != self.dst:
                if own is None:
                    return
                # the header (shapes: known now; a small host-built tensor of its own) travels at once; the payload sends
                # are queued behind the event, i.e. behind this item's forward.  dst can therefore read the header, post its
                # receives and go on launching its next forward without waiting for anybody's kernels.
                hdr = self._header(payload).to(self.wire)
                keep.extend([hdr] + payload)
                works.append(dist.isend(hdr, self.dst, group=self.group))
                if ev is not None:
                    self._side.wait_event(ev)
                    for t in payload:
                        t.record_stream(self._side)
                works.extend(dist.batch_isend_irecv([dist.P2POp(dist.isend, t, self.dst, self.group) for t in payload]))
                return
            if own is not None:
                items.append((index, list(own)))
            # ALL headers first (they are already on their way: peers send them before their kernels finish), THEN one
            # grouped post of every payload receive: the receives of different peers run concurrently, each on its own
            # xGMI link, and nothing this rank has to read on the host is queued behind a payload
            peers = [p for p in range(self.world) if p != self.dst and self._has_item(rnd, p)]
            headers = []
            for peer in peers:
                hdr = torch.zeros(self.n_tensors, _MAXD + 2, dtype=torch.int64, device=self.wire)
                dist.recv(hdr, src=peer, group=self.group)
                headers.append(hdr)
            ops = []
            for peer, hdr in zip(peers, headers):
                h = hdr.cpu()
                bufs = []
                for b in range(self.n_tensors):
                    nd = int(h[b, 1])
                    buf = torch.empty(tuple(int(v) for v in h[b, 2:2 + nd]), dtype=_DTYPES[int(h[b, 0])], device=self.wire)
                    ops.append(dist.P2POp(dist.irecv, buf, peer, self.group))
                    bufs.append(buf)
                    self.bytes_received += buf.numel() * buf.element_size()
                items.append((rnd * self.world + peer, bufs))
            if ops:
                works.extend(dist.batch_isend_irecv(ops))

        if self._side is not None:
            with torch.cuda.stream(self._side):
                comm()
        else:
            comm()
        self._inflight = (rnd, works, keep, items)

    def stats(self) -> Dict[str, float]:
        """Gather diagnostics of this rank: bytes received, host seconds blocked waiting for transfers, and (device wire)
        the milliseconds the compute stream stood waiting for the transfer stream -- the part of the gather that was not
        overlapped with the next forward.  Synchronises the device."""
        if self._wait_events:
            torch.cuda.synchronize(self.wire)
            self._fold_wait_events(keep_pending=False)
        exposed_ms = self._wait_ms_folded
        return {"bytes_received": float(self.bytes_received), "host_wait_s": self.host_wait_s,
                "stream_wait_ms": exposed_ms, "rounds": float(self._round)}

    def _fold_wait_events(self, keep_pending: bool) -> None:
        """Add the elapsed time of recorded (start, stop) pairs to the running total and drop them; with keep_pending only the
        pairs whose stop event has completed (no synchronisation), else all (the caller has synchronised)."""
        rest = []
        for a, b in self._wait_events:
            if keep_pending and not b.query():
                rest.append((a, b))
            else:
                self._wait_ms_folded += a.elapsed_time(b)
        if keep_pending and len(rest) > 64:                            # nothing completes (a stalled stream): wait for the oldest
            a, b = rest.pop(0)
            b.synchronize()
            self._wait_ms_folded += a.elapsed_time(b)
        self._wait_events = rest

    # ------------------------------------------------------------------ API
    def submit(self, index: int, tensors: Sequence[torch.Tensor]) -> None:
        if index != self._round * self.world + self.rank or index >= self.num_items:
            raise ValueError(f"rank {self.rank} submitted item {index} in round {self._round}: items must arrive in "
                             f"ascending order of shard_indices({self.num_items}, {self.rank}, {self.world})")
        if len(tensors) != self.n_tensors:
            raise ValueError(f"expected {self.n_tensors} tensors per item, got {len(tensors)}")
        self._retire()
        self._post_round(self._round, tensors)
        self._round += 1

    def finish(self) -> None:
        """Drains: a rank whose shard is shorter (or empty) still takes part in the remaining rounds on `dst`."""
        mine = len(shard_indices(self.num_items, self.rank, self.world))
        if self._round != mine:
            raise RuntimeError(f"rank {self.rank} submitted {self._round} of its {mine} items before finish()")
        if self.rank == self.dst:
            while self._round < self.num_rounds:           # rounds in which dst has no item of its own
                self._retire()
                self._post_round(self._round, None)
                self._round += 1
        self._retire()


def gather_arrays(local: Dict[int, Sequence[torch.Tensor]], num_items: int, dst: int = 0,
                  group: Optional[dist.ProcessGroup] = None, device: Optional[torch.device] = None,
                  tensors_per_item: Optional[int] = None) -> Optional[Dict[int, List[torch.Tensor]]]:
    """Gather per-item tensor tuples to `dst` in one call (small jobs and tests; the CLI streams through
    RoundGatherer directly so that only one round is ever resident).

    `local` maps the global item index (as assigned by shard_indices) to that item's tensors; shapes may differ per
    item.  Returns {index: [tensors]} on dst, None elsewhere.  `tensors_per_item` must be given when some rank may own
    nothing (it cannot be inferred there)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = shard_indices(num_items, rank, world)
    assert sorted(local) == mine, f"rank {rank} holds {sorted(local)} but owns {mine}"
    if tensors_per_item is None:
        if not local:
            raise ValueError("tensors_per_item is required on a rank that owns no item")
        tensors_per_item = len(next(iter(local.values())))
    out: Dict[int, List[torch.Tensor]] = {}
    if device is None and local:
        device = next(iter(local.values()))[0].device
    g = RoundGatherer(num_items, on_item=lambda i, ts: out.__setitem__(i, list(ts)), tensors_per_item=tensors_per_item,
                      dst=dst, group=group, device=device)
    for idx in mine:
        g.submit(idx, local[idx])
    g.finish()
    return out if rank == dst else None
