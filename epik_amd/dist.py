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
    # EPIK_AMD_DIST_WORLD1=1: a process group of ONE rank, so that the collectives of the multi-GPU paths (their dtypes,
    # split sizes, streams) run on the real backend -- RCCL -- on a box with a single device (tests, bench.py)
    if world <= 1 and os.environ.get("EPIK_AMD_DIST_WORLD1") != "1":
        return None
    import torch
    import torch.distributed as dist
    if world <= 1:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_PORT", "29577")
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


AMB_NONE = 0xFFFFFFFF  # amb_order of a branch no ambiguous key of the shard reached


def amb_slots(seqs: np.ndarray, seq_offsets: np.ndarray, char_class: np.ndarray, world: int):
    """Slots for the reads that may hold an ambiguous k-mer (any character that is not one plain
    state), laid out for the exchange of `place_kmer_sharded`: the reads of owner j (`owner_bounds`)
    get the slots j * per_owner .. in read order.  Returns (int32 amb_slot[n] with -1 = none, per_owner);
    identical on every rank, as all of them hold the same batch."""
    n = len(seq_offsets) - 1
    plain = np.array([bin(int(c)).count("1") == 1 for c in char_class], dtype=bool)
    dirty = (~plain[np.asarray(seqs, dtype=np.uint8)]).astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(dirty)])
    offs = np.asarray(seq_offsets, dtype=np.int64)
    has = (csum[offs[1:]] - csum[offs[:-1]]) > 0
    slot = np.full(n, -1, dtype=np.int32)
    counts = []
    for j in range(world):
        b, e = owner_bounds(n, j, world)
        counts.append(int(has[b:e].sum()))
    per_owner = max(counts) if counts else 0
    for j in range(world):
        b, e = owner_bounds(n, j, world)
        idx = np.nonzero(has[b:e])[0] + b
        slot[idx] = j * per_owner + np.arange(len(idx), dtype=np.int32)
    return slot, per_owner


def combine_amb(order, avg):
    """[shards, rows, N] records of the shards -> per branch the average probability of the ambiguous
    key of smallest order over the shards, 0 where no shard has one (place.cpp:385-388: only the first
    ambiguous key that reaches a branch scores it -- first over the whole database)."""
    import torch
    order = order.to(torch.int64) & 0xFFFFFFFF
    best = order.min(dim=0).values
    chosen = (order == best.unsqueeze(0)) & (order != AMB_NONE)
    # a key lives in exactly one shard, so at most one shard is chosen per branch
    return torch.where(chosen, avg, torch.zeros_like(avg)).sum(dim=0)


def place_kmer_sharded(accumulate: Callable, finish: Callable, n_reads: int, dist, gather_to: int | None = 0,
                       amb_slot: np.ndarray | None = None, amb_per_owner: int = 0):
    """K-mer-space-sharded placement of one batch that every rank holds in full.

    accumulate(n_rows_padded, amb_slot, amb_rows) -> (scores float32 [n_rows_padded, N], counts int16
        [n_rows_padded, N] (the bits of uint16 counts), amb_order int32 [amb_rows, N] (the bits of uint32
        orders), amb_avg float32 [amb_rows, N]) -- the last two None when amb_rows == 0 --
        torch tensors on this rank's device: the raw per-branch sums of THIS rank's lists for reads
        0..n_reads-1 (the padding rows behind n_reads must be zero) and the records of their ambiguous
        k-mers (Placer.accumulate_device on a placer created with shard_index=rank, shard_count=world);
    finish(begin, end, scores, counts, amb_slot, amb_avg) -> (rows, n_rows, kmer_counts) numpy arrays for
        the reads [begin, end) this rank owns, from their totals; amb_slot (int32 [end - begin], -1 = none)
        indexes the rows of amb_avg (Placer.finish_device).
    amb_slot / amb_per_owner: from `amb_slots` (None: no read of the batch can hold an ambiguous k-mer).

    Returns like `place_sharded`: the rank's own rows, or everything on `gather_to`.
    """
    rank, _, world = env_rank_world()
    if amb_slot is None or amb_per_owner == 0:
        amb_slot, amb_per_owner = None, 0
    if dist is None:
        scores, counts, _, amb_avg = accumulate(n_reads, amb_slot, amb_per_owner)
        if amb_avg is not None:  # one shard: the record is the total; AMB_NONE rows carry avg 0 already
            pass
        return finish(0, n_reads, scores, counts, amb_slot, amb_avg)
    import torch
    per = -(-n_reads // world)
    scores, counts, amb_order, amb_avg = accumulate(per * world, amb_slot, amb_per_owner * world)
    assert scores.shape[0] == per * world and counts.shape == scores.shape
    begin, end = owner_bounds(n_reads, rank, world)

    def exchange(part):
        # slice j of `part` goes to rank j; what arrives is every rank's partial of MY slice.  The rows
        # cross as bytes: the collectives of some backends do not take 16-bit integers.
        sent = part.contiguous().view(torch.uint8)
        received = torch.empty_like(sent)
        dist.all_to_all_single(received, sent)
        return received.view(part.dtype).view(world, part.shape[0] // world, -1)

    totals = []
    for part in (scores, counts):
        received = exchange(part)
        total = received[0].clone()
        for g in range(1, world):  # fixed order: the float32 sums do not depend on the transport
            total += received[g]
        totals.append(total)
    my_slot = my_avg = None
    if amb_per_owner:
        my_avg = combine_amb(exchange(amb_order), exchange(amb_avg))
        my_slot = amb_slot[begin:end].copy()
        my_slot[my_slot >= 0] -= rank * amb_per_owner
    mine = finish(begin, end, totals[0][:end - begin], totals[1][:end - begin], my_slot, my_avg)
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
    seq_offsets = np.ascontiguousarray(seq_offsets, dtype=np.uint64)
    d_seqs = torch.from_numpy(np.ascontiguousarray(seqs, dtype=np.uint8)).to(device)
    d_offs = torch.from_numpy(seq_offsets.view(np.int64)).to(device)
    stream = torch.cuda.current_stream(device).cuda_stream
    N, keep = placer.num_branches, placer.keep_at_most
    # the count width the batch needs: the device entry points do not choose it themselves
    placer.choose_counts(int(np.diff(seq_offsets.astype(np.int64)).max()) if n else 0)

    def accumulate(n_rows_padded, amb_slot, amb_rows):
        scores = torch.zeros((n_rows_padded, N), dtype=torch.float32, device=device)
        counts = torch.zeros((n_rows_padded, N), dtype=torch.int16, device=device)
        d_slot = order = avg = None
        if amb_rows:
            d_slot = torch.from_numpy(np.ascontiguousarray(amb_slot, dtype=np.int32)).to(device)
            order = torch.full((amb_rows, N), -1, dtype=torch.int32, device=device)  # AMB_NONE
            avg = torch.zeros((amb_rows, N), dtype=torch.float32, device=device)
        placer.accumulate_device(d_seqs.data_ptr(), d_offs.data_ptr(), n, scores.data_ptr(), counts.data_ptr(), stream,
                                 d_amb_slot=d_slot.data_ptr() if amb_rows else 0,
                                 d_amb_order=order.data_ptr() if amb_rows else 0,
                                 d_amb_avg=avg.data_ptr() if amb_rows else 0)
        torch.cuda.synchronize(device)
        out = (scores, counts, order, avg)
        return tuple(None if x is None else x.cpu() for x in out) if host_staging else out

    def finish(begin, end, scores, counts, amb_slot, amb_avg):
        m = end - begin
        rows = np.zeros((m, keep), dtype=capi.PLACEMENT)
        n_rows = np.zeros(m, dtype=np.uint32)
        kmer_counts = np.zeros((m, keep), dtype=np.uint32)
        if m == 0:
            return rows, n_rows, kmer_counts
        scores, counts = scores.to(device).contiguous(), counts.to(device).contiguous()
        d_slot = d_avg = None
        if amb_avg is not None and amb_slot is not None:
            d_slot = torch.from_numpy(np.ascontiguousarray(amb_slot, dtype=np.int32)).to(device)
            d_avg = amb_avg.to(device).contiguous()
        d_rows = torch.zeros(m * keep * 2, dtype=torch.float64, device=device)   # 16 B per row
        d_n_rows = torch.zeros(m, dtype=torch.int32, device=device)
        d_kc = torch.zeros(m * keep, dtype=torch.int32, device=device)
        # lengths come from the offsets of the owned reads (absolute offsets are fine: only differences are used)
        placer.finish_device(d_offs.data_ptr() + 8 * begin, m, scores.data_ptr(), counts.data_ptr(),
                             d_rows.data_ptr(), d_n_rows.data_ptr(), d_kc.data_ptr(), stream,
                             d_amb_slot=d_slot.data_ptr() if d_slot is not None else 0,
                             d_amb_avg=d_avg.data_ptr() if d_avg is not None else 0)
        torch.cuda.synchronize(device)
        rows[:] = d_rows.cpu().numpy().view(capi.PLACEMENT).reshape(m, keep)
        n_rows[:] = d_n_rows.cpu().numpy().view(np.uint32)
        kmer_counts[:] = d_kc.cpu().numpy().view(np.uint32).reshape(m, keep)
        return rows, n_rows, kmer_counts

    return accumulate, finish


# ---------------------------------------------------------------------------------------------------
# K-mer-space shard with partial LISTS (include/epik_amd.h, "partial LISTS"): a shard's lists reach a
# small part of a large tree per read, so what crosses is, per read and slice of the branch range, only
# the rows that received a k-mer -- kilobytes per read where the dense vectors are 6 bytes per branch --
# and the exchange of batch b runs under the accumulate of batch b + 1.
#
# An ENGINE is the two kernel halves behind a small protocol (`ListsGpuEngine` below: the HIP placer;
# tests/dist_worker_kmer.py has a numpy one for the gloo tests on CPU):
#     slices, entry_bytes                          geometry of the lists, the same on every rank
#     compute_stream, comm_stream                  torch streams, or None (no device)
#     begin(seqs, seq_offsets) -> batch            uploads a batch every rank holds in full
#     accumulate(batch, n_parts, amb_slot, amb_rows, min_entries=0) -> Partials   (asynchronous)
#     finish(batch, begin, end, entries, index, amb_slot, amb_avg) -> (rows, n_rows, kmer_counts)
#         entries / index: one tensor per shard, in shard order -- the part's entries as that shard
#         wrote them and its [end - begin (padded), slices, 2] index
# ---------------------------------------------------------------------------------------------------
class Partials:
    """What one accumulate leaves: `entries` (uint8, the parts one after the other), `index` (int32
    [per * n_parts, slices, 2] = {first entry inside the part, entries}), `part_entries` (int64 [n_parts],
    how many entries each part takes in `entries`), the ambiguous records (or None), the capacity of
    `entries` in entries, and the event after which all of it is there (None: already)."""

    def __init__(self, entries, index, part_entries, amb_order, amb_avg, cap, done=None, entry_bytes=None):
        self.entries, self.index, self.part_entries = entries, index, part_entries
        self.amb_order, self.amb_avg, self.cap, self.done = amb_order, amb_avg, cap, done
        # bytes per entry of THESE lists (8, or 16 with 32-bit counts): the width is chosen per batch, and the next
        # batch's accumulate has chosen again by the time this one crosses and finishes (None: the engine's)
        self.entry_bytes = entry_bytes


class ListsGpuEngine:
    """The accumulate_lists / finish_lists halves of a `Placer` created with shard_index / shard_count
    (a large tree: `placer.partial_info()["lists"]`).  `host_staging`: the exchange runs on host
    tensors (a gloo group) -- the lists are copied out after accumulate and back in for finish."""

    def __init__(self, placer, device, host_staging: bool = False, overlap: bool = True):
        import torch
        info = placer.partial_info()
        if not info["lists"]:
            raise ValueError("this placer's kernels leave dense partial vectors (a small tree): use place_kmer_sharded")
        self.placer, self.device, self.host_staging = placer, device, host_staging
        self.slices, self.num_branches, self.keep = info["slices"], info["num_branches"], placer.keep_at_most
        self.entry_bytes = info["entry_bytes"]
        self.postings_per_kmer = info["postings_per_kmer"]
        self.kmer_size = placer.kmer_size
        self.compute_stream = torch.cuda.current_stream(device)
        self.comm_stream = torch.cuda.Stream(device) if (overlap and not host_staging) else None
        self.margin = 1.3  # room in `entries` over the estimate; grows when a batch overflowed

    def begin(self, seqs: np.ndarray, seq_offsets: np.ndarray):
        import torch
        seq_offsets = np.ascontiguousarray(seq_offsets, dtype=np.uint64)
        n = len(seq_offsets) - 1
        lengths = np.diff(seq_offsets.astype(np.int64)) if n else np.zeros(0, np.int64)
        batch = {"n": n, "offsets": seq_offsets,
                 "d_seqs": torch.from_numpy(np.ascontiguousarray(seqs, dtype=np.uint8)).to(self.device),
                 "d_offs": torch.from_numpy(seq_offsets.view(np.int64)).to(self.device),
                 "longest": int(lengths.max()) if n else 0,
                 "kmers": int(np.maximum(lengths - (self.kmer_size - 1), 0).sum())}
        return batch

    def accumulate(self, batch, n_parts: int, amb_slot, amb_rows: int, min_entries: int = 0):
        import torch
        n, S, dev = batch["n"], self.slices, self.device
        per = -(-n // n_parts) if n else 0
        self.placer.choose_counts(batch["longest"])  # the count width (and with it the entry format) of this batch
        self.entry_bytes = self.placer.partial_info()["entry_bytes"]
        cap = max(int(batch["kmers"] * self.postings_per_kmer * self.margin) + 4096, int(min_entries))
        cap = min(cap, (1 << 32) - 1)
        with torch.cuda.stream(self.compute_stream):
            entries = torch.empty(cap * self.entry_bytes, dtype=torch.uint8, device=dev)
            index = torch.zeros((per * n_parts, S, 2), dtype=torch.int32, device=dev)
            part_entries = torch.zeros(n_parts, dtype=torch.int64, device=dev)
            d_slot = order = avg = None
            if amb_rows:
                d_slot = torch.from_numpy(np.ascontiguousarray(amb_slot, dtype=np.int32)).to(dev)
                order = torch.full((amb_rows, self.num_branches), -1, dtype=torch.int32, device=dev)  # AMB_NONE
                avg = torch.zeros((amb_rows, self.num_branches), dtype=torch.float32, device=dev)
            self.placer.accumulate_lists_device(
                batch["d_seqs"].data_ptr(), batch["d_offs"].data_ptr(), n, n_parts, entries.data_ptr(), cap,
                index.data_ptr(), part_entries.data_ptr(), self.compute_stream.cuda_stream,
                d_amb_slot=d_slot.data_ptr() if amb_rows else 0, d_amb_order=order.data_ptr() if amb_rows else 0,
                d_amb_avg=avg.data_ptr() if amb_rows else 0)
            done = self.compute_stream.record_event()
        batch["_keep"] = (d_slot,)  # (alive until the kernels have run)
        return Partials(entries, index, part_entries, order, avg, cap, done, entry_bytes=self.entry_bytes)

    def finish(self, batch, begin: int, end: int, entries, index, amb_slot, amb_avg):
        import torch
        from . import capi
        m, keep, dev = end - begin, self.keep, self.device
        # the count width (and entry format) this batch was accumulated with: a later batch's accumulate may have
        # chosen another one in between (a short-read batch in front of a batch with a read of more than 255 k-mers)
        self.placer.choose_counts(batch["longest"])
        rows = np.zeros((m, keep), dtype=capi.PLACEMENT)
        n_rows = np.zeros(m, dtype=np.uint32)
        kmer_counts = np.zeros((m, keep), dtype=np.uint32)
        if m == 0:
            return rows, n_rows, kmer_counts
        with torch.cuda.stream(self.compute_stream):
            entries = [e.to(dev) for e in entries]
            index = [x.to(dev).contiguous() for x in index]
            d_slot = d_avg = None
            if amb_avg is not None and amb_slot is not None:
                d_slot = torch.from_numpy(np.ascontiguousarray(amb_slot, dtype=np.int32)).to(dev)
                d_avg = amb_avg.to(dev).contiguous()
            d_rows = torch.zeros(m * keep * 2, dtype=torch.float64, device=dev)  # 16 B per row
            d_n_rows = torch.zeros(m, dtype=torch.int32, device=dev)
            d_kc = torch.zeros(m * keep, dtype=torch.int32, device=dev)
            self.placer.finish_lists_device(
                batch["d_offs"].data_ptr() + 8 * begin, m, [e.data_ptr() if e.numel() else 0 for e in entries],
                [x.data_ptr() for x in index], d_rows.data_ptr(), d_n_rows.data_ptr(), d_kc.data_ptr(),
                self.compute_stream.cuda_stream, d_amb_slot=d_slot.data_ptr() if d_slot is not None else 0,
                d_amb_avg=d_avg.data_ptr() if d_avg is not None else 0)
            self.compute_stream.synchronize()
            rows[:] = d_rows.cpu().numpy().view(capi.PLACEMENT).reshape(m, keep)
            n_rows[:] = d_n_rows.cpu().numpy().view(np.uint32)
            kmer_counts[:] = d_kc.cpu().numpy().view(np.uint32).reshape(m, keep)
        return rows, n_rows, kmer_counts


def _exchange_lists(engine, parts: Partials, n_reads: int, dist, rank: int, world: int):
    """One exchange step: every rank sends each peer the part of that peer's reads -- its entries (as
    many as the part takes: all-to-all with split sizes, known after a gather of the part sizes) and its
    index (equal splits) -- one link per peer.  Returns (entries[g], index[g]) in shard order for the
    reads this rank owns, and the total this rank's parts took (for the overflow check)."""
    import torch
    eb = parts.entry_bytes if getattr(parts, "entry_bytes", None) else engine.entry_bytes
    S = engine.slices
    per = -(-n_reads // world) if n_reads else 0
    staged = getattr(engine, "host_staging", False)
    sizes_mine = parts.part_entries.cpu() if staged else parts.part_entries
    # the part sizes AND the room this rank had: every rank must know whether ANY rank's lists found no room before
    # the first data collective -- a rank that overflowed alone and left here by itself would meet the others'
    # all-to-all with another collective (capacities differ per shard: so does what overflows)
    # (... and the width of its entries: the finisher reads every shard's list in ONE format -- ranks whose
    # choose_counts differ, devices of two kinds or a width forced on one, would mis-size the all-to-all and mis-read
    # what arrives; shard_place.hip makes its handles agree on the widest, here the disagreement is an error every rank sees)
    mine = torch.cat([sizes_mine.to(torch.int64).reshape(-1),
                      torch.tensor([int(parts.cap), int(eb)], dtype=torch.int64, device=sizes_mine.device)]).contiguous()
    gathered = torch.empty(world * (world + 2), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(gathered, mine)
    gathered = gathered.cpu().view(world, world + 2)
    sizes, caps, widths = gathered[:, :world], gathered[:, world], gathered[:, world + 1]   # sizes[g][r]: entries shard g holds for the reads of rank r
    if bool((widths != eb).any()):
        raise RuntimeError(f"k-mer-space shard: the ranks' partial lists have entries of {sorted(set(int(w) for w in widths))} bytes "
                           "(different count widths: force one with epik_amd_placer_set_wide_counts on every rank)")
    mine_total = int(sizes[rank].sum())
    if bool((sizes.sum(dim=1) > caps).any()):         # every rank leaves together; the caller repeats the accumulate
        return None, None, mine_total
    send_split = [int(x) * eb for x in sizes[rank]]
    recv_split = [int(sizes[g][rank]) * eb for g in range(world)]
    send = parts.entries[:mine_total * eb]
    send_index = parts.index
    if staged:
        send, send_index = send.cpu(), send_index.cpu()
    recv = torch.empty(sum(recv_split), dtype=torch.uint8, device=send.device)
    dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=recv_split, input_split_sizes=send_split)
    recv_index = torch.empty_like(send_index)
    # (the index crosses as bytes: the collectives of some backends do not take every integer type)
    dist.all_to_all_single(recv_index.view(torch.uint8).view(-1), send_index.contiguous().view(torch.uint8).view(-1))
    entries, at = [], 0
    for g in range(world):
        entries.append(recv[at:at + recv_split[g]])
        at += recv_split[g]
    index = [recv_index.view(world, per, S, 2)[g] for g in range(world)]
    return entries, index, mine_total


def place_kmer_sharded_lists(engine, batches, dist, char_class=None, gather_to: int | None = 0):
    """K-mer-space-sharded placement of a sequence of batches, each held in full by every rank, with
    partial lists and the exchange of a batch overlapped with the accumulate of the next one.

    batches: iterable of (seqs uint8, seq_offsets uint64); char_class: the 256-entry class table (None: no
    read holds an ambiguous character, no slots are made).  Yields, per batch and in order, what
    `place_sharded` returns: the rank's own rows (gather_to None), or everything on `gather_to`.
    """
    import contextlib
    import torch
    rank, _, world = env_rank_world()
    if dist is None:
        rank, world = 0, 1
    comm = getattr(engine, "comm_stream", None)
    compute = getattr(engine, "compute_stream", None)

    def start(seqs, offs):
        n = len(offs) - 1
        slot, per_owner = (None, 0)
        if char_class is not None:
            slot, per_owner = amb_slots(seqs, offs, char_class, world)
            if per_owner == 0:
                slot = None
        batch = engine.begin(seqs, offs)
        parts = engine.accumulate(batch, world, slot, per_owner * world)
        return {"n": n, "batch": batch, "parts": parts, "slot": slot, "per_owner": per_owner}

    def complete(job):
        n, batch, parts = job["n"], job["batch"], job["parts"]
        begin, end = owner_bounds(n, rank, world)
        slot, per_owner = job["slot"], job["per_owner"]
        my_slot = my_avg = None
        if dist is None:
            if parts.done is not None:
                parts.done.synchronize()
            total = int(parts.part_entries.sum())
            while total > parts.cap:   # more room, once more (the estimate learns)
                engine.margin *= 1.5
                parts = engine.accumulate(batch, world, slot, per_owner * world, min_entries=total)
                if parts.done is not None:
                    parts.done.synchronize()
                total = int(parts.part_entries.sum())
            # (the width of THIS batch's entries: the engine's may be the next batch's by now, whose accumulate is under way)
            entries, index = [parts.entries[:total * (parts.entry_bytes or engine.entry_bytes)]], [parts.index]
            if per_owner:
                my_slot, my_avg = slot, parts.amb_avg
            return engine.finish(batch, 0, n, entries, index, my_slot, my_avg)
        while True:
            ctx = torch.cuda.stream(comm) if comm is not None else contextlib.nullcontext()
            with ctx:
                if parts.done is not None:
                    if comm is not None:
                        comm.wait_event(parts.done)   # the exchange waits for THIS batch's accumulate only
                    else:
                        parts.done.synchronize()
                entries, index, total = _exchange_lists(engine, parts, n, dist, rank, world)
                again = entries is None   # (the same on every rank: decided from the gathered sizes and capacities)
                if not again and per_owner:
                    def cross(x):
                        x = x.cpu() if getattr(engine, "host_staging", False) else x
                        sent = x.contiguous().view(torch.uint8)
                        got = torch.empty_like(sent)
                        dist.all_to_all_single(got.view(-1), sent.view(-1))
                        return got.view(x.dtype).view(world, x.shape[0] // world, -1)
                    my_avg = combine_amb(cross(parts.amb_order), cross(parts.amb_avg))
                    my_slot = slot[begin:end].copy()
                    my_slot[my_slot >= 0] -= rank * per_owner
                done = comm.record_event() if comm is not None else None
            if not again:
                break
            engine.margin *= 1.5
            # (a rank whose own lists had room keeps its capacity: only what overflowed grows)
            parts = engine.accumulate(batch, world, slot, per_owner * world, min_entries=max(total or 0, 1))
        if done is not None and compute is not None:
            compute.wait_event(done)
        mine = engine.finish(batch, begin, end, entries, [x[:max(end - begin, 0)] for x in index], my_slot, my_avg)
        if gather_to is None:
            return mine
        gathered = [None] * world if rank == gather_to else None
        dist.gather_object(mine, gathered, dst=gather_to)
        if rank != gather_to:
            return None
        return tuple(np.concatenate([p[i] for p in gathered], axis=0) for i in range(3))

    pending = None
    for seqs, offs in batches:
        job = start(seqs, offs)          # batch b + 1 accumulates ...
        if pending is not None:
            yield complete(pending)      # ... while batch b crosses and finishes
        pending = job
    if pending is not None:
        yield complete(pending)
