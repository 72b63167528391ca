"""Multi-GPU glue: one process per GPU, reads sharded across ranks, database replicated.

The placement path shards by reads (each read is placed independently, place.cpp:230-268;
the database is read-only), so there is NO data-path collective: every rank places its own
contiguous shard.  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU) is used
only for rendezvous, barriers, the max-over-ranks time of bench.py and, when a caller wants
the whole result in one place, a gather of the small per-read rows.
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
