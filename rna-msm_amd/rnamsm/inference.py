"""extract_feat: the reference CLI's per-RNA loop (RNA_MSM_Inference.py:90-168) on the HIP path.

For every id in `data.MSA_list` (sorted), reads `<root>/<MSA_path>/<id>.a2m_msa2`, runs the 10-layer forward and
writes `<id>_emb.npy` float32 (L, 768) and `<id>_atp.npy` float32 (120, L, L) -- the NPY v1 C-order files the
downstream SS / RSA predictors consume.  With torch.distributed initialised (one process per GPU),
ids are sharded round-robin over ranks and either written by the rank that computed them (default) or gathered
to rank 0 over RCCL first (`gather_to_rank0=True`).
"""
from __future__ import annotations

import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops, sharding
from .alphabet import RNAAlphabet
from .config import Config
from .model import MSATransformer
from .msa import load_msa_tokens


def load_checkpoint(model: MSATransformer, path: str, device) -> None:
    """torch.load(path)['state_dict'] with strict key matching (RNA_MSM_Inference.py:133-135)."""
    blob = torch.load(path, map_location="cpu", weights_only=False)
    state = blob["state_dict"] if isinstance(blob, dict) and "state_dict" in blob else blob
    model.load_state_dict(state, strict=True)


def find_msa_files(msa_dir: Path, ids: List[str]) -> Dict[str, Path]:
    """`*.a2m_msa2` files whose stem (up to the first '.') is a requested id (dataset.py:46-64)."""
    if not msa_dir.exists():
        raise FileNotFoundError(msa_dir)
    if not msa_dir.is_dir():
        raise NotADirectoryError(msa_dir)
    wanted = set(ids)
    if not wanted:
        raise ValueError("Passed an empty split file set")
    found = {f.stem.split(".")[0]: f for f in sorted(msa_dir.glob("*.a2m_msa2")) if f.stem.split(".")[0] in wanted}
    if len(found) != len(wanted):
        raise FileNotFoundError(f"{len(wanted) - len(found)} specified split files not found in directory")
    return found


def crop_tokens(tokens: np.ndarray, max_seqlen: int, rng: np.random.RandomState) -> np.ndarray:
    """RandomCropDataset.__getitem__ (dataset.py:142-158): when the alignment (incl. <cls>) is wider than
    max_seqlen keep <cls> plus a random window of max_seqlen - 1 columns."""
    seqlen = tokens.shape[-1]
    if seqlen <= max_seqlen:
        return tokens
    start = rng.randint(1, seqlen - max_seqlen) if seqlen - max_seqlen > 1 else 1
    return np.concatenate([tokens[..., :1], tokens[..., start:start + max_seqlen - 1]], -1)


class _AsyncNpyWriter:
    """Device tensors -> `.npy` files off the critical path: the D2H copy runs on a side stream into pinned buffers and a
    worker thread waits for it and calls np.save, so the GPU starts the next MSA while this one's 126 MB (L = 512) of maps
    are still on their way to disk.  At most `depth` MSAs wait in the queue and `workers` are being written (bounds the pinned memory).  `workers` threads share
    the queue (np.save releases the GIL while it writes: a list of small alignments is bound by the writer, 442 MB for 64 of
    them); an alignment's two files are written by one worker, alignments may finish out of order.  `close()` drains the
    queue and re-raises a worker exception."""

    def __init__(self, device: torch.device, depth: int = 2, workers: int = 2):
        self._stream = torch.cuda.Stream(device)
        self._q: "queue.Queue" = queue.Queue(maxsize=depth)
        self._err: Optional[BaseException] = None
        self._threads = [threading.Thread(target=self._run, name=f"rnamsm-npy-writer-{i}", daemon=True) for i in range(workers)]
        for t in self._threads:
            t.start()

    def _run(self) -> None:
        while True:
            item = self._q.get()
            if item is None:
                return
            event, jobs, done = item
            try:
                if self._err is None:
                    event.synchronize()
                    for path, host in jobs:
                        np.save(path, host.numpy())
                    done()
            except BaseException as e:                  # noqa: BLE001  (reported by close())
                self._err = e

    def submit(self, jobs, done, after: Optional[torch.cuda.Event] = None) -> None:
        """jobs: [(path, device tensor)], written in this order; done(): called by the worker after the last file.
        after: an event recorded behind the kernels that produced the tensors -- the copies then wait for THAT point of the compute
        stream only, not for what was enqueued since (the next packed group's forward: the CLI's pipelined pool); None = for
        everything enqueued so far."""
        if after is not None:
            self._stream.wait_event(after)
        else:
            self._stream.wait_stream(torch.cuda.current_stream())
        staged = []
        with torch.cuda.stream(self._stream):
            for path, t in jobs:
                host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                host.copy_(t, non_blocking=True)
                t.record_stream(self._stream)           # the allocator must not recycle t before the copy has run
                staged.append((path, host))
            event = torch.cuda.Event()
            event.record(self._stream)
        self._q.put((event, staged, done))              # blocks when `depth` MSAs are already in flight

    def close(self) -> None:
        for _ in self._threads:
            self._q.put(None)
        for t in self._threads:
            t.join()
        if self._err is not None:
            raise self._err


# data.batch_small_msas: which alignments wait for company, and when a group is full.  An alignment of up to SMALL_MSA_TOKENS
# tokens is "small"; a group's frame -- members x max rows x max columns -- stays within FRAME_TOKENS and within twice the real
# tokens, and holds at most GROUP_MEMBERS alignments.  (The limit was 1536 tokens while groups were formed in list order: from
# ~2 k tokens on the padding of an ill-matched frame cost more than the shared launches returned, tools/ragged_batch_timing.py.
# With groups formed by shape -- plan_groups -- frames are tight and 3072 measured x1.82 against x1.54 on a list of 4-24 rows x
# 40-120 columns, tools/cli_throughput.py.)
# (Frames: 16 k tokens / 32 members until the pool existed; a batch of 64 alignments of 8 x 64 runs at 676 k residues/s against
# 649 k for 32 -- GEMM tile quantisation shrinks with the token count, tools/small_batch_knobs.py -- hence 32 k / 64.)
SMALL_MSA_TOKENS, FRAME_TOKENS, GROUP_MEMBERS = 3072, 32768, 64
# ... and within FRAME_MAP_ELEMS = members x (max columns)^2: the maps (row_attn and atp: layers x heads x C^2 floats per member,
# 480 B per element at 10 x 12) scale with C^2, not with the tokens -- 32 shallow alignments of 1 x 1024 would be 16 GB of maps for
# a frame of 32 k tokens, and batching gains nothing at that C.  2^20 elements = 0.5 GB of atp per group: 64 members up to C = 128,
# 16 at C = 256, none above C = 1024 / alone.
FRAME_MAP_ELEMS = 1 << 20


def joins_group(group_shapes: List[tuple], shape: tuple) -> bool:
    """Does a small alignment of `shape` (rows, columns) join the waiting group, or must the group be flushed first?"""
    if len(group_shapes) >= GROUP_MEMBERS:
        return False
    trial = list(group_shapes) + [shape]
    cols = max(s[1] for s in trial)
    frame = len(trial) * max(s[0] for s in trial) * cols
    return frame <= FRAME_TOKENS and frame <= 2 * sum(s[0] * s[1] for s in trial) and len(trial) * cols * cols <= FRAME_MAP_ELEMS


# Without a gather the order in which alignments are computed is free (every alignment has its own files), so the small ones
# of the list wait in a POOL (at most POOL_MSAS of them) and are grouped by shape instead of by list position: sorted by
# (rows, columns) a frame's members are alike, groups fill up to GROUP_MEMBERS / FRAME_TOKENS instead of stopping at the first
# alignment that would pad too much, and a large alignment in between no longer flushes anything.  64 alignments of 2-12 rows
# x 40-80 columns: 12 groups in list order, 4 sorted (tools/cli_throughput.py).
POOL_MSAS = 256


# what one more launch set costs, in tokens of frame (~5.5 ms at ~600 k tokens/s): the padding one more member may add to a frame.
# (Half and a quarter of it -- tighter frames, more groups -- measured the same within 5 % on lists of 64 and 256 alignments.)
GROUP_OVERHEAD_TOKENS = 3300


def _greedy_groups(shapes: List[tuple], order: List[int]) -> List[List[int]]:
    """Fill groups in the given order; an alignment joins the waiting group while joins_group allows it AND the padding it adds
    to the frame costs less than a launch set of its own (GROUP_OVERHEAD_TOKENS)."""
    groups: List[List[int]] = []
    cur: List[int] = []
    rows = cols = 0
    for j in order:
        r, c = shapes[j]
        if cur:
            grown = (len(cur) + 1) * max(rows, r) * max(cols, c) - len(cur) * rows * cols       # what the frame grows by
            if not joins_group([shapes[i] for i in cur], shapes[j]) or grown - r * c > GROUP_OVERHEAD_TOKENS:
                groups.append(sorted(cur))
                cur, rows, cols = [], 0, 0
        cur.append(j)
        rows, cols = max(rows, r), max(cols, c)
    if cur:
        groups.append(sorted(cur))
    return groups


def _clustered_groups(shapes: List[tuple]) -> List[List[int]]:
    """Seed a group with the largest alignment left, then keep adding the one that pads the frame least (joins_group's rule,
    evaluated for all candidates at once) while that padding costs less than a launch set of its own."""
    rc = np.asarray(shapes, dtype=np.int64).reshape(-1, 2)
    tok = rc[:, 0] * rc[:, 1]
    left = np.ones(len(shapes), dtype=bool)
    groups: List[List[int]] = []
    while left.any():
        seed = int(np.flatnonzero(left)[np.argmax(tok[left])])         # (the first of equals)
        left[seed] = False
        g, rows, cols, real = [seed], int(rc[seed, 0]), int(rc[seed, 1]), int(tok[seed])
        while len(g) < GROUP_MEMBERS and left.any():
            frame = (len(g) + 1) * np.maximum(rows, rc[:, 0]) * np.maximum(cols, rc[:, 1])
            ok = left & (frame <= FRAME_TOKENS) & (frame <= 2 * (real + tok))
            ok &= (len(g) + 1) * np.maximum(cols, rc[:, 1]) ** 2 <= FRAME_MAP_ELEMS
            added_padding = frame - len(g) * rows * cols - tok
            ok &= added_padding <= GROUP_OVERHEAD_TOKENS
            if not ok.any():
                break
            best = int(np.argmin(np.where(ok, added_padding, np.iinfo(np.int64).max)))
            left[best] = False
            g.append(best)
            rows, cols, real = max(rows, int(rc[best, 0])), max(cols, int(rc[best, 1])), real + int(tok[best])
        groups.append(sorted(g))
    return groups


def frame_tokens(shapes: List[tuple], groups: List[List[int]]) -> int:
    return sum(len(g) * max(shapes[j][0] for j in g) * max(shapes[j][1] for j in g) for g in groups)


def plan_groups(shapes: List[tuple]) -> List[List[int]]:
    """Partition pooled small alignments (`shapes` = (rows, columns) each) into ragged-batch groups; returns lists of
    positions into `shapes`, every position exactly once, members of a group in ascending position.  Two plans are made -- fill
    greedily in (rows, columns) order; cluster around the largest alignment left -- and the one with the smaller cost (tokens
    of frame, padding included, plus GROUP_OVERHEAD_TOKENS per group) is taken, the first on a tie.  Lists of 64 / 256
    alignments of 2-12 rows x 40-80 columns: 3 / 7 groups, x3.6 / x4.6 against one by one; of 4-24 x 40-120: 6 / 15 groups, x1.9 /
    x2.3 (tools/cli_throughput.py)."""
    if not shapes:
        return []
    plans = [_greedy_groups(shapes, sorted(range(len(shapes)), key=lambda j: (shapes[j][0], shapes[j][1], j))),
             _clustered_groups(shapes)]
    cost = [frame_tokens(shapes, p) + GROUP_OVERHEAD_TOKENS * len(p) for p in plans]
    return plans[0] if cost[0] <= cost[1] else plans[1]


# ---- token-packed groups (exact mode, data.pack_small_msas; round 4): forward_ragged then runs rnamsm_forward_packed, which
# pads nothing, so a group is bounded by what it really holds -- PACKED_TOKENS tokens, PACKED_MEMBERS members, FRAME_MAP_ELEMS map
# elements (sum of C^2) -- and shapes need not match: groups are cut in list order (the gather path keeps its order for free).
# Measured (tools/packed_batch_timing.py, 4-24 rows x 41-121 columns): 4 k tokens 361 k residues/s, 10 k 522 k, 21 k 633 k, 34 k 621 k,
# 49 k 678 k, 66 k 689 k (GEMM tile quantisation and the folded LayerNorm's threshold) against 147 k one by one -- hence 64 k
# tokens / 256 members; and alignments of up to PACKED_SMALL_TOKENS wait for company (16-64 rows x 60-200 columns, ~5 k tokens
# each: x1.4-1.5 packed against alone; framed they lost).
# (round 5: 8192 -> 16384 -- 64 alignments of 16-64 rows x 60-200 columns: 96.2 -> 100.1 MSA/s, f16x3 172.5 -> 180.8, bf16 neutral and
# left at 8192; tools/cli_throughput.py with PACKED_SMALL / PACKED_SMALL_16)
# (PACKED_TOKENS 65536 -> 131072: a 77 k-token list of small alignments as one group instead of two: 385 -> 408 MSA/s)
PACKED_TOKENS, PACKED_MEMBERS, PACKED_SMALL_TOKENS = 131072, 256, 16384
# The CLI's pooled path pipelines its token-packed groups (group g+1 enqueued before group g's error word is read); a pool that would be
# one group of at least this many tokens is dealt into two so that there is something to overlap (RNAMSM_PIPELINE_SPLIT_TOKENS: A/B, 0 = never)
PIPELINE_SPLIT_TOKENS = int(os.environ.get("RNAMSM_PIPELINE_SPLIT_TOKENS", "24576"))
# In a 16-bit arithmetic mode the packed batch runs in that mode too since round 5 (rnamsm_forward_packed: every Linear on the
# 16-bit matrix cores, attention on the exact descriptor kernels), so the same limit applies.  (Round 4 had sent those small
# alignments through the EXACT packed path, with limits of 1024 / 2048 tokens: there was no 16-bit packed batch.)
PACKED_SMALL_TOKENS_16BIT = {"bf16": 8192, "f16x3": 16384}


def r_c_below(shape: tuple, limit: int) -> bool:
    return shape[0] * shape[1] < limit


def joins_packed(group_shapes: List[tuple], shape: tuple) -> bool:
    """Does a small alignment of `shape` join the waiting token-packed group?"""
    trial = list(group_shapes) + [shape]
    return (len(trial) <= PACKED_MEMBERS and sum(r * c for r, c in trial) <= PACKED_TOKENS
            and sum(c * c for _, c in trial) <= FRAME_MAP_ELEMS)


def plan_packed_groups(shapes: List[tuple], fold_min_tokens: int = 0, split_tokens: int = 0, _min_groups: int = 1) -> List[List[int]]:
    """Partition pooled small alignments into token-packed groups, in list order, every position exactly once.  The number of
    groups is the least the three bounds allow; members are then dealt so that the groups hold about the same number of tokens
    (64 + 6 alignments would otherwise run as one full group and one nearly empty launch set).
    fold_min_tokens > 0 (exact mode): alignments of at least that many tokens take the folded LayerNorm, smaller ones do not, and a
    launch set is one or the other -- the two classes are planned separately, so that no group has to be run as two batches
    (MSATransformer.forward_packed would split a mixed one: correct, but two launch sets).
    split_tokens > 0: a pool that would run as ONE group IN ALL, of at least that many tokens, is dealt into two, so that the caller
    has something to pipeline (the second group's forward over the first one's deliveries: +7 % on 64 small alignments; a pool of
    several groups pipelines already, and splitting its smaller class measured -2 %: profiles/r06_cli_pipeline_ab.log)."""
    if not shapes:
        return []
    if split_tokens > 0:
        plan = plan_packed_groups(shapes, fold_min_tokens)
        if len(plan) == 1 and len(shapes) >= 2 and sum(r * c for r, c in shapes) >= split_tokens:
            plan = plan_packed_groups(shapes, 0, 0, 2)
        return plan
    if fold_min_tokens > 0:
        big = [j for j, (r, c) in enumerate(shapes) if r * c >= fold_min_tokens]
        if 0 < len(big) < len(shapes):
            small = [j for j in range(len(shapes)) if r_c_below(shapes[j], fold_min_tokens)]
            return ([[big[j] for j in g] for g in plan_packed_groups([shapes[j] for j in big])]
                    + [[small[j] for j in g] for g in plan_packed_groups([shapes[j] for j in small])])
    tok = [r * c for r, c in shapes]
    k = max(-(-len(shapes) // PACKED_MEMBERS), -(-sum(tok) // PACKED_TOKENS), -(-sum(c * c for _, c in shapes) // FRAME_MAP_ELEMS), _min_groups)
    while True:
        # next-fit with a soft budget of 1/k of the tokens: a group closes once it has reached the budget (the member that crosses
        # it still joins, bounds permitting), or earlier when the next member would break a hard bound
        budget = -(-sum(tok) // k)
        groups: List[List[int]] = [[]]
        t = 0
        for j in range(len(shapes)):
            if groups[-1] and (t >= budget or not joins_packed([shapes[i] for i in groups[-1]], shapes[j])):
                groups.append([])
                t = 0
            groups[-1].append(j)
            t += tok[j]
        if len(groups) <= k:
            return groups
        k = len(groups)             # the hard bounds cut earlier than planned: aim for that many, evenly filled


def extract_feat(cfg: Config, model: Optional[MSATransformer] = None, gather_to_rank0: bool = False,
                 async_io: bool = True) -> List[str]:
    """async_io: read/tokenise the next alignment on a helper thread while the GPU runs the current one, and move the
    outputs to disk through `_AsyncNpyWriter`; False = the reference's strictly sequential loop (same files)."""
    device = torch.device(cfg.data.device)
    if device.type != "cuda":
        raise RuntimeError("this build runs on the MI355X HIP path only (data.device=cuda); there is no CPU path")
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if device.index is None:          # "cuda": this rank's GPU under a launcher, else the process's current device
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank)) if world > 1 else torch.cuda.current_device())
    alphabet = RNAAlphabet.from_architecture(cfg.data.architecture)
    if rank == 0:
        print(f"Maximum Number of MSA Seqs:{cfg.data.max_seqs_per_msa}")
        print(f"Inference on: {device}")

    root = Path(cfg.data.root_path)
    with open(root / cfg.data.MSA_list) as f:
        ids = sorted(f.read().splitlines())
    ids = [i for i in ids if i]
    files = find_msa_files(root / cfg.data.MSA_path, ids)

    if model is None:
        model = MSATransformer(alphabet, embed_dim=cfg.model.embed_dim, num_attention_heads=cfg.model.num_attention_heads,
                               num_layers=cfg.model.num_layers, embed_positions_msa=cfg.model.embed_positions_msa,
                               dropout=cfg.model.dropout, attention_dropout=cfg.model.attention_dropout,
                               activation_dropout=cfg.model.activation_dropout, max_tokens_per_msa=cfg.data.max_tokens,
                               max_seqlen=cfg.data.max_seqlen)
        load_checkpoint(model, cfg.data.model_path, device)
    model = model.eval().to(device)
    model.gemm_dtype = cfg.model.gemm_dtype

    save_dir = root / cfg.data.MSA_path
    save_dir.mkdir(parents=True, exist_ok=True)
    rng = np.random.RandomState(42)
    mine = sharding.shard_indices(len(ids), rank, world)
    written: List[str] = []
    # the library launches on the calling thread's current device: make it this rank's GPU on the main thread and on
    # the reader thread (whose greedy sub-sampling runs on the device)
    torch.cuda.set_device(device)

    def write(rna_id: str, emb: np.ndarray, atp: np.ndarray) -> None:
        np.save(save_dir / f"{rna_id}_atp.npy", atp)
        np.save(save_dir / f"{rna_id}_emb.npy", emb)
        written.append(rna_id)

    def read(idx: int) -> np.ndarray:
        """Reader + row sub-sampling + crop of one alignment.  All randomness of the CLI lives here, drawn from ONE stream
        in the reference's order -- per item `sample-pretrained`'s weighted draw (dataset.py:88-90), then the crop
        (dataset.py:147) -- seeded 42 like the reference's global numpy state (RNA_MSM_Inference.py:17), so a single
        process reproduces the reference's choices; under sharding every rank draws for its own items."""
        torch.cuda.set_device(device)
        tokens = load_msa_tokens(files[ids[idx]], alphabet, cfg.data.max_seqs_per_msa, cfg.data.sample_method, device=device,
                                 rng=rng)
        return crop_tokens(tokens, cfg.data.max_seqlen, rng)

    gathering = gather_to_rank0 and world > 1
    writer = _AsyncNpyWriter(device) if async_io and (not gathering or rank == 0) else None
    reader = ThreadPoolExecutor(1, thread_name_prefix="rnamsm-msa-reader") if async_io else None

    def emit(rna_id: str, emb: torch.Tensor, atp: torch.Tensor, after: Optional[torch.cuda.Event] = None) -> None:
        if writer is not None:
            writer.submit([(save_dir / f"{rna_id}_atp.npy", atp), (save_dir / f"{rna_id}_emb.npy", emb)],
                          lambda r=rna_id: written.append(r), after=after)
        else:
            write(rna_id, emb.cpu().numpy(), atp.cpu().numpy())

    # gather_to_rank0: outputs travel to rank 0 one ROUND (one MSA per rank) at a time, round k's RCCL transfers
    # overlapping round k+1's forward, and are handed to the writer as they arrive -- at most one round is resident
    # Transport: RCCL point-to-point where it comes up; otherwise the ranks agree (over the default gloo group) to stage the same
    # gather through host memory -- the outputs arrive on rank 0 either way (sharding.negotiate_gather_transport).  A default group
    # that is not gloo (a caller's own nccl world) is used as it is, as before.
    gather_group = None
    if gathering and dist.get_backend() == "gloo":
        gather_group, label, _ = sharding.negotiate_gather_transport(device, want_backend=os.environ.get("RNAMSM_GATHER_BACKEND", "nccl"))
        if gather_group is False:
            raise RuntimeError(f"gather_to_rank0: no transport to rank 0 ({label}); run without RNAMSM_GATHER_TO_RANK0 "
                               f"(every rank then writes its own files)")
        if label != "primary" and rank == 0:
            print(f"gather to rank 0: {label}")
    gatherer = sharding.RoundGatherer(len(ids), on_item=lambda i, ts: emit(ids[i], ts[0], ts[1]), tensors_per_item=2,
                                      dst=0, device=device, group=gather_group) if gathering else None
    try:
        with torch.no_grad():
            pending = reader.submit(read, mine[0]) if reader and len(mine) else None
            def deliver(idx: int, emb: torch.Tensor, atp: torch.Tensor, after: Optional[torch.cuda.Event] = None) -> None:
                if gatherer is not None:
                    gatherer.submit(idx, (emb, atp))
                else:
                    emit(ids[idx], emb, atp, after)

            # data.batch_small_msas: small alignments go through ONE launch set per group (forward_ragged: padded into one
            # frame, every MSA scaled by its own depth); a lone forward of a few hundred tokens costs 5.5 ms on a mostly
            # idle chip.  Groups: by shape from the pool (plan_groups); under gather_to_rank0 the RoundGatherer needs this
            # rank's items in list order, so there only CONSECUTIVE small alignments share a group.
            # (16-bit modes: only with data.batch_small_msas_16bit -- a batch's token count selects the GEMM kernels there, so
            # an alignment's files would depend on its neighbours in the list at the mode's rounding level)
            exact = model.gemm_dtype == "f32"
            framed16 = bool(getattr(cfg.data, "batch_small_msas_16bit", False)) and ops.get_param("attn16") != 0
            # groups are token-packed (no frame, no padding -- rnamsm_forward_packed, in the model's arithmetic mode), so any small
            # alignments share one -- unless 16-bit framed batches were asked for (data.batch_small_msas_16bit)
            packing = (bool(getattr(cfg.data, "batch_small_msas", True)) and bool(getattr(cfg.data, "pack_small_msas", True))
                       and (exact or not framed16))
            batching = bool(getattr(cfg.data, "batch_small_msas", True)) and (exact or framed16 or packing)
            pooled = batching and gatherer is None
            small_limit = (PACKED_SMALL_TOKENS if exact else PACKED_SMALL_TOKENS_16BIT.get(model.gemm_dtype, 1024)) if packing else SMALL_MSA_TOKENS
            group: List = []                                          # (idx, tokens on the device)
            pool: List = []                                           # (idx, tokens on the host)
            inflight: List = [None]                                   # the pipelined pool's group whose results are still on the device

            def finish_packed(entry) -> None:
                """Second half of a pipelined packed group: its error word and outputs, read behind the event recorded after its launches."""
                members_, (res, mode), ev = entry
                toks_ = [t for _, t in members_]
                try:
                    outs = model.forward_ragged_finish(toks_, res, mode, after=ev)
                    if outs is not res:
                        ev = None                                 # a rerun: its outputs are behind everything enqueued so far
                    if outs is None:                              # <pad> inside the batch: the framed rerun of forward_ragged
                        outs = model.forward_ragged(toks_, packed=False)
                except IndexError:                                # name the offending alignment: one by one
                    outs, ev = [alone(i, t) for i, t in members_], None
                if not all(o["emb"].is_contiguous() and o["atp"].is_contiguous() for o in outs):
                    ev = None                                     # .contiguous() below would launch copies on the compute stream
                for (i, _), out in zip(members_, outs):
                    deliver(i, out["emb"].contiguous(), out["atp"].contiguous(), ev)

            def finish_one(entry) -> None:
                """Second half of a pipelined lone forward (the one-by-one loop of alignments too large to wait for company)."""
                _, idx_, t_dev, has_pad, out, ev = entry
                done = model.finish_forward_one(t_dev, out, has_pad, need_repr=False, what=ids[idx_], after=ev)
                deliver(idx_, done["emb"], done["atp"], ev if done is out else None)

            def finish_any(entry) -> None:
                (finish_one if entry[0] == "one" else finish_packed)(entry)

            def settle() -> None:
                """Read back and deliver what is still in flight, if anything: a packed group or a lone forward."""
                if inflight[0] is not None:
                    entry, inflight[0] = inflight[0], None
                    finish_any(entry)

            def run_pool() -> None:
                shapes_ = [tuple(t.shape) for _, t in pool]
                fold_min = ops.get_param("ln_fold_min_tokens") if (exact and model.fold_layernorm and ops.get_param("ln_fold") == 1) else 0
                if not packing:
                    for members in plan_groups(shapes_):
                        group.extend((pool[j][0], torch.from_numpy(pool[j][1]).to(device)) for j in members)
                        flush()
                    pool.clear()
                    return
                # Token-packed groups, PIPELINED (round 6): group g+1 is enqueued before group g's error word is read, so g's
                # device-to-host copies and file writes run under g+1's forward instead of after it with the GPU idle (64 small
                # alignments used to be one blocking 117 ms forward followed by 26 ms of deliveries).  A pool that would be ONE group
                # of at least PIPELINE_SPLIT_TOKENS tokens is dealt into two for that.  Same launch sets per group as before: every
                # member's files are byte for byte those of its own forward (exact mode).
                # (one stream carries every forward: what belongs to group g -- its error word, its outputs -- is read behind an EVENT
                # recorded right after g's launches, on side streams, or it would wait for g+1's forward as well)
                # The last group of a pool STAYS in flight when run_pool returns (inflight[0]): the reader goes on, the next pool's first group
                # is enqueued behind it, and only then is it read back; settle() ends the pipeline (before a forward outside the pool, at the
                # end of the list).
                waiting = inflight[0]                                 # (members' (idx, tokens), begun forward, event) of the group in flight
                inflight[0] = None

                try:
                    for members in plan_packed_groups(shapes_, fold_min, split_tokens=PIPELINE_SPLIT_TOKENS):
                        members_ = [(pool[j][0], torch.from_numpy(pool[j][1]).to(device)) for j in members]
                        if len(members_) == 1:
                            if waiting is not None:
                                finish_any(waiting)
                                waiting = None
                            group.extend(members_)
                            flush()
                            continue
                        begun = model.forward_ragged_begin([t for _, t in members_])
                        ev = torch.cuda.Event()
                        ev.record(torch.cuda.current_stream())
                        if waiting is not None:
                            finish_any(waiting)
                        waiting = (members_, begun, ev)
                finally:
                    inflight[0] = waiting                             # also when a group raised: what is in flight is still delivered (salvage)
                pool.clear()

            def alone(i: int, t: torch.Tensor) -> dict:
                """One member of a group by itself, in the arithmetic the GROUP runs in (what a small alignment's
                files hold must not depend on whether it had company).  Exact mode: its own forward IS the packed batch's
                arithmetic, bit for bit.  16-bit modes: a packed batch's attention runs on the exact kernels, a lone forward's on
                the 16-bit ones -- so there a lone small alignment is a packed batch of one."""
                if packing and not exact:
                    try:
                        return model.forward_ragged([t], packed=True)[0]
                    except IndexError:
                        raise IndexError(f"{ids[i]}: token or position index out of range") from None
                return model.checked_forward_one(t, need_repr=False, what=ids[i])

            def flush() -> None:
                if len(group) == 1:
                    idx0, t0 = group[0]
                    out = alone(idx0, t0)
                    deliver(idx0, out["emb"].contiguous(), out["atp"].contiguous())
                elif group:
                    try:
                        outs = model.forward_ragged([t for _, t in group], packed=packing)
                    except IndexError:                                # name the offending alignment: one by one
                        outs = [alone(i, t) for i, t in group]
                    for (i, _), out in zip(group, outs):
                        deliver(i, out["emb"].contiguous(), out["atp"].contiguous())
                group.clear()

            def read_and_run() -> None:
                nonlocal pending
                for n, idx in enumerate(mine):
                    rna_id = ids[idx]
                    tokens = pending.result() if reader else read(idx)
                    if reader and n + 1 < len(mine):
                        pending = reader.submit(read, mine[n + 1])       # parsed while the GPU runs this MSA
                    if (batching and tokens.size <= small_limit and 2 * tokens.shape[1] ** 2 <= FRAME_MAP_ELEMS
                            and not (tokens == alphabet.padding_idx).any()):
                        if pooled:
                            pool.append((idx, tokens))
                            # (packing: a full group's worth of tokens starts NOW -- the reader goes on under its forward -- instead of
                            # waiting for POOL_MSAS alignments or the end of the list)
                            if len(pool) >= POOL_MSAS or (packing and sum(t.size for _, t in pool) >= PACKED_TOKENS):
                                run_pool()
                            continue
                        if not (joins_packed if packing else joins_group)([tuple(t.shape) for _, t in group], tuple(tokens.shape)):
                            flush()                                       # the group is full (framed: this one would pad too much)
                        group.append((idx, torch.from_numpy(tokens).to(device)))
                        continue
                    flush()
                    has_pad = bool((tokens == alphabet.padding_idx).any())          # on the host: no device round trip before the launches
                    t_dev = torch.from_numpy(tokens).to(device)
                    if gatherer is not None:       # (the gather orders its transfers behind the compute stream: one by one there, as before)
                        settle()
                        out = model.checked_forward_one(t_dev, has_pad, need_repr=False, what=rna_id)   # emb + atp are all that is written
                        deliver(idx, out["emb"], out["atp"])
                        continue
                    # PIPELINED (round 6): this forward is enqueued BEFORE the previous one's error word and outputs are read -- the ~3 ms
                    # of host work per alignment (weight-table check, ~140 launches, delivery) no longer idle the GPU between forwards
                    out = model.forward_one(t_dev, has_pad, need_repr=False)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())
                    settle()
                    inflight[0] = ("one", idx, t_dev, has_pad, out, ev)

            try:
                read_and_run()
            except Exception:
                # a bad alignment late in the list must not cost the results already computed or read: what waits in the
                # group / pool is run and written (as the one-by-one loop would have done before reaching it), then the
                # error goes up.  KeyboardInterrupt / SystemExit are not caught: they propagate at once.  Under a gatherer the
                # peers are not known to be in the same state (a deliver() could block on them): no salvage there.  A second
                # failure in here is reported and dropped in favour of the first.
                if gatherer is None:
                    try:
                        flush()
                        run_pool()
                        settle()
                    except Exception as second:                       # noqa: BLE001
                        import warnings
                        warnings.warn(f"while writing the results computed before the error: {type(second).__name__}: {second}")
                raise
            flush()
            run_pool()
            settle()
            if gatherer is not None:
                gatherer.finish()
    finally:
        if reader is not None:
            reader.shutdown(wait=True)
        if writer is not None:
            writer.close()
    if rank == 0:
        print(f"Done! Generated files are saved at {save_dir}")
    position = {rna_id: n for n, rna_id in enumerate(ids)}
    return sorted(written, key=position.__getitem__)                  # in list order, whatever order the groups ran in
