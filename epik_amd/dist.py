"""Multi-GPU glue: one process per GPU.

Common case -- reads sharded across ranks, database replicated: the placement path shards by
reads (each read is placed independently, place.cpp:230-268; the database is read-only), so
there is NO data-path collective: every rank places its own contiguous shard
(`place_sharded`).  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU) is used
only for rendezvous, barriers, the max-over-ranks time of bench.py and, when a caller wants
the whole result in one place, a gather of the small per-read rows.

Database larger than one GPU (BASELINE configs[4]) -- `place_kmer_sharded`: rank g
holds the posting lists of the k-mer codes with code % G == g, every rank accumulates ALL reads
of a batch against its lists, and the per-read [num_branches] score/count vectors are summed over
the ranks with ONE exchange step: a direct all-to-all (each GPU sends every peer the slice of
reads that peer owns -- one xGMI link per peer, no ring) followed by a local sum in rank order,
i.e. a reduce-scatter over the read dimension with a fixed summation order.
"""
from __future__ import annotations

import os
from typing import Callable, Tuple

import numpy as np


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced shard [begin, end) of `n_items` for `rank` of `world`."""
    return n_items * rank // world, n_items * (rank + 1) // world


def shard_reads(seqs: np.ndarray, seq_offsets: np.ndarray, rank: int, world: int):
    """The rank's contiguous slice of a packed read batch, re-based to offset 0."""
    n = len(seq_offsets) - 1
    b, e = shard_bounds(n, rank, world)
    lo, hi = int(seq_offsets[b]), int(seq_offsets[e])
    return seqs[lo:hi], (seq_offsets[b:e + 1] - seq_offsets[b]).astype(np.uint64), (b, e)


def init_process_group(backend: str | None = None, device_index: int | None = None):
    """Initialises torch.distributed from the environment; returns the module or None when
    WORLD_SIZE == 1.  backend None = "nccl" (RCCL) with a GPU, else "gloo"."""
    rank, local_rank, world = env_rank_world()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {}
    if backend == "nccl":
        kwargs["device_id"] = torch.device("cuda", local_rank if device_index is None else device_index)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, **kwargs)
    return dist


def max_over_ranks(value: float, dist, device=None) -> float:
    """MAX all-reduce of one float (the bench's whole-job time)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def place_sharded(place_fn: Callable, seqs: np.ndarray, seq_offsets: np.ndarray, dist, gather_to: int | None = 0):
    """Every rank places its shard with `place_fn(seqs, offsets) -> (rows, n_rows, counts)`;
    with `gather_to` the shards are concatenated, in read order, on that rank (others get None)."""
    rank, _, world = env_rank_world()
    if dist is None:
        return place_fn(seqs, seq_offsets)
    my_seqs, my_offs, _ = shard_reads(seqs, seq_offsets, rank, world)
    mine = place_fn(my_seqs, my_offs)
    if gather_to is None:
        return mine
    parts = [None] * world if rank == gather_to else None
    dist.gather_object(mine, parts, dst=gather_to)
    if rank != gather_to:
        return None
    return tuple(np.concatenate([p[i] for p in parts], axis=0) for i in range(3))


def owner_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Equal-sized slices (the last ones may be short or empty): what the all-to-all of
    `place_kmer_sharded` needs.  Slice length = ceil(n_items / world)."""
    per = -(-n_items // world) if n_items else 0
    return min(n_items, rank * per), min(n_items, (rank + 1) * per)


def place_kmer_sharded(accumulate: Callable, finish: Callable, n_reads: int, dist, gather_to: int | None = 0):
    """K-mer-space-sharded placement of one batch that every rank holds in full.

    accumulate(n_rows_padded) -> (scores float32 [n_rows_padded, N], counts int32 [n_rows_padded, N])
        torch tensors on this rank's device: the raw per-branch sums of THIS rank's lists for reads
        0..n_reads-1; the padding rows behind n_reads must be zero
        (Placer.accumulate_device on a placer created with shard_index=rank, shard_count=world);
    finish(begin, end, scores, counts) -> (rows, n_rows, kmer_counts) numpy arrays for the reads
        [begin, end) this rank owns, from their totals (Placer.finish_device).

    Returns like `place_sharded`: the rank's own rows, or everything on `gather_to`.
    """
    rank, _, world = env_rank_world()
    if dist is None:
        scores, counts = accumulate(n_reads)
        return finish(0, n_reads, scores, counts)
    import torch
    per = -(-n_reads // world)
    scores, counts = accumulate(per * world)
    assert scores.shape[0] == per * world and counts.shape == scores.shape
    begin, end = owner_bounds(n_reads, rank, world)
    totals = []
    for part in (scores, counts):
        # slice j of `part` goes to rank j; what arrives is every rank's partial of MY slice
        received = torch.empty_like(part)
        dist.all_to_all_single(received, part.contiguous())
        received = received.view(world, per, -1)
        total = received[0].clone()
        for g in range(1, world):  # fixed order: the float32 sums do not depend on the transport
            total += received[g]
        totals.append(total)
    mine = finish(begin, end, totals[0][:end - begin], totals[1][:end - begin])
    if gather_to is None:
        return mine
    parts = [None] * world if rank == gather_to else None
    dist.gather_object(mine, parts, dst=gather_to)
    if rank != gather_to:
        return None
    return tuple(np.concatenate([p[i] for p in parts], axis=0) for i in range(3))


def kmer_sharded_gpu_fns(placer, seqs: np.ndarray, seq_offsets: np.ndarray, device, host_staging: bool = False):
    """(accumulate, finish) for `place_kmer_sharded` over a `Placer` created with
    shard_index / shard_count, for one batch given as host arrays.  The exchange runs on device
    tensors (RCCL) unless `host_staging` (a gloo group: the partial vectors cross in host memory)."""
    import torch
    from . import capi
    n = len(seq_offsets) - 1
    d_seqs = torch.from_numpy(np.ascontiguousarray(seqs, dtype=np.uint8)).to(device)
    d_offs = torch.from_numpy(np.ascontiguousarray(seq_offsets, dtype=np.uint64).view(np.int64)).to(device)
    stream = torch.cuda.current_stream(device).cuda_stream
    N, keep = placer.num_branches, placer.keep_at_most

    def accumulate(n_rows_padded):
        scores = torch.zeros((n_rows_padded, N), dtype=torch.float32, device=device)
        counts = torch.zeros((n_rows_padded, N), dtype=torch.int32, device=device)
        placer.accumulate_device(d_seqs.data_ptr(), d_offs.data_ptr(), n, scores.data_ptr(), counts.data_ptr(), stream)
        torch.cuda.synchronize(device)
        return (scores.cpu(), counts.cpu()) if host_staging else (scores, counts)

    def finish(begin, end, scores, counts):
        m = end - begin
        rows = np.zeros((m, keep), dtype=capi.PLACEMENT)
        n_rows = np.zeros(m, dtype=np.uint32)
        kmer_counts = np.zeros((m, keep), dtype=np.uint32)
        if m == 0:
            return rows, n_rows, kmer_counts
        scores, counts = scores.to(device).contiguous(), counts.to(device).contiguous()
        d_rows = torch.zeros(m * keep * 2, dtype=torch.float64, device=device)   # 16 B per row
        d_n_rows = torch.zeros(m, dtype=torch.int32, device=device)
        d_kc = torch.zeros(m * keep, dtype=torch.int32, device=device)
        # lengths come from the offsets of the owned reads (absolute offsets are fine: only differences are used)
        placer.finish_device(d_offs.data_ptr() + 8 * begin, m, scores.data_ptr(), counts.data_ptr(),
                             d_rows.data_ptr(), d_n_rows.data_ptr(), d_kc.data_ptr(), stream)
        torch.cuda.synchronize(device)
        rows[:] = d_rows.cpu().numpy().view(capi.PLACEMENT).reshape(m, keep)
        n_rows[:] = d_n_rows.cpu().numpy().view(np.uint32)
        kmer_counts[:] = d_kc.cpu().numpy().view(np.uint32).reshape(m, keep)
        return rows, n_rows, kmer_counts

    return accumulate, finish
