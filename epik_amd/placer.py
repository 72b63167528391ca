"""Host-side mirror of `epik::placer` (reference epik/include/epik/place.h:81-140)
over the C-ABI of libepik_amd.so.

Same constructor arguments and `place()` contract as the reference class:

    placer(db, tree, keep_at_most, keep_factor, max_threads)      place.h:94-95
    placed_collection place(seq_records, num_threads)              place.h:103

* the constructor precomputes the pendant lengths (place.cpp:99-125) and uploads
  the database to HBM through `epik_amd_placer_create`;
* `place()` groups identical sequences (place.cpp:73-81, 207-212), sends the unique
  reads through the boundary, and joins distal/pendant lengths onto the returned
  rows (place.cpp:435-437).

The heavy lifting is the HIP kernel; nothing here computes a score.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Iterable, List, Sequence, Tuple

import numpy as np

from . import alphabet, capi


@dataclass
class Placement:
    """`epik::impl::placement` (place.h:45-56)."""

    branch_id: int
    score: float          # float32 value
    weight_ratio: float   # double
    count: int
    distal_length: float
    pendant_length: float


@dataclass
class PlacedSequence:
    """`epik::impl::placed_sequence` (place.h:59-68)."""

    sequence: str
    placements: List[Placement]


@dataclass
class PlacedCollection:
    """`epik::impl::placed_collection` (place.h:72-75): sequence -> headers, plus
    one PlacedSequence per unique sequence (first-occurrence order; the reference's
    order is std::unordered_map iteration order, place.cpp:57-61)."""

    sequence_map: dict
    placed_seqs: List[PlacedSequence]


def pendant_lengths(branch_length: np.ndarray, subtree_num_nodes: np.ndarray,
                    subtree_total_length: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(distal, pendant) per post-order id, place.cpp:99-125 / :435."""
    branch_length = np.asarray(branch_length, dtype=np.float64)
    num = np.asarray(subtree_num_nodes, dtype=np.float64)
    tot = np.asarray(subtree_total_length, dtype=np.float64)
    distal = branch_length / 2                      # :110
    mean = np.where(num > 1, tot / np.maximum(num, 1), 0.0)  # :117-121
    return distal, mean + distal                    # :123


def make_desc(offsets, values, *, states: str, kmer_size: int, num_branches: int, threshold, log_threshold=None,
              keep_at_most: int = 7, keep_factor: float = 0.01, device: int = 0, char_class=None, keys=None,
              sparse: bool = False, holds_shard=None):
    """`epik_amd_placer_desc` over host arrays (uint32 / uint64 offsets are handed over as they are, no
    copy).  `keys` (ascending uint32 codes that have a list): the sparse form, `offsets` then has
    len(keys) + 1 entries; `sparse=True` turns dense offsets into that form first (EPIK_AMD_SPARSE_DESC=1
    does it for every descriptor: the GPU tests run on both forms).  `holds_shard=(g, G)`: the arrays hold shard g of
    G of a database already (only the lists of the codes with code % G == g: `epik_amd_placer_desc.shard`).
    Returns (desc, the arrays it points into -- keep them alive as long as the descriptor)."""
    sigma = alphabet.alphabet_size(states)
    if log_threshold is None:
        log_threshold = alphabet.log_threshold(np.float32(threshold))
    offsets = np.asarray(offsets)
    num_keys = sigma ** int(kmer_size)
    if keys is None and (sparse or os.environ.get("EPIK_AMD_SPARSE_DESC") == "1"):
        lens = np.diff(offsets.astype(np.int64))
        present = np.nonzero(lens)[0]
        keys = present.astype(np.uint32)
        offsets = np.concatenate([[0], np.cumsum(lens[present])]).astype(np.uint64)
    if keys is not None:
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        assert offsets.shape[0] == keys.shape[0] + 1
    num_entries = int(offsets[-1])
    if offsets.dtype == np.uint64 and offsets.flags.c_contiguous:
        off, bits = offsets, 64
    elif num_entries <= 0xFFFFFFFF:
        off, bits = np.ascontiguousarray(offsets, dtype=np.uint32), 32
    else:
        off, bits = np.ascontiguousarray(offsets, dtype=np.uint64), 64
    vals = np.ascontiguousarray(values)
    if vals.dtype.itemsize != 8:
        raise ValueError("values must be 8-byte {uint32 branch, float32 score} records")
    cls = np.ascontiguousarray(
        alphabet.char_class_table(states) if char_class is None else char_class, dtype=np.uint32)
    desc = capi.PlacerDesc(
        abi_version=capi.ABI_VERSION, kmer_size=int(kmer_size), alphabet_size=sigma,
        num_branches=int(num_branches), keep_at_most=int(keep_at_most), offset_bits=bits,
        keep_factor=float(keep_factor), threshold=float(threshold),
        log_threshold=float(log_threshold), num_keys=int(num_keys),
        num_entries=num_entries, offsets=off.ctypes.data, values=vals.ctypes.data,
        char_class=cls.ctypes.data, device=int(device),
        shard=0 if holds_shard is None or int(holds_shard[1]) <= 1 else (int(holds_shard[0]) | int(holds_shard[1]) << 16),
        keys=keys.ctypes.data if keys is not None else None, num_present=int(keys.shape[0]) if keys is not None else 0)
    return desc, (off, vals, cls, keys)


def plan(db, *, shard_index: int = 0, shard_count: int = 1, free_bytes: int = 288 << 30, **kw) -> capi.Plan:
    """`epik_amd_placer_plan`: kernel, layout and device-image sizes create() would choose for a
    synthetic / loaded database `db` -- no device needed."""
    kw.setdefault("keys", getattr(db, "keys", None))
    kw.setdefault("holds_shard", getattr(db, "shard", None))
    desc, keep = make_desc(db.offsets, db.values, states=db.states, kmer_size=db.kmer_size,
                           num_branches=db.num_branches, threshold=db.threshold, log_threshold=db.log_threshold, **kw)
    out = capi.Plan()
    capi.check(capi.load().epik_amd_placer_plan(ctypes.byref(desc), shard_index, shard_count, int(free_bytes),
                                                ctypes.byref(out)))
    del keep
    return out


def plan_sizes(*, states: str, kmer_size: int, num_branches: int, bins, keep_at_most: int = 7, shard_index: int = 0,
               shard_count: int = 1, free_bytes: int = 288 << 30) -> capi.Plan:
    """`epik_amd_placer_plan_sizes`: the plan of a database known by its SIZES only -- `bins` = (length, lists[,
    lists_in_runs]) per length of posting list the placer (its shard) keeps.  No device, no postings."""
    arr = (capi.ListBin * max(len(bins), 1))()
    for i, b in enumerate(bins):
        arr[i].length, arr[i].lists = int(b[0]), int(b[1])
        arr[i].lists_in_runs = int(b[2]) if len(b) > 2 else 0
    out = capi.Plan()
    capi.check(capi.load().epik_amd_placer_plan_sizes(int(kmer_size), alphabet.alphabet_size(states), int(num_branches),
                                                      int(keep_at_most), arr, len(bins), int(shard_index), int(shard_count),
                                                      int(free_bytes), ctypes.byref(out)))
    return out


def list_bins(db, shard_index: int = 0, shard_count: int = 1):
    """The histogram `plan_sizes` takes, of a database at hand (tests; capacity reports of a loaded database)."""
    offsets = np.asarray(db.offsets).astype(np.int64)
    lens = np.diff(offsets)
    codes = np.asarray(db.keys, dtype=np.int64) if getattr(db, "keys", None) is not None else np.arange(len(lens))
    mine = (codes % shard_count == shard_index) & (lens > 0)
    br = db.values["branch"].astype(np.int64)
    steps_ok = np.ones(len(br), dtype=bool)
    steps_ok[1:] = np.diff(br) == 1
    broken = np.zeros(len(br) + 1, dtype=np.int64)   # (how many positions of a list other than its first break the run)
    inner = np.ones(len(br), dtype=bool)
    inner[offsets[:-1][lens > 0]] = False
    broken[1:] = np.cumsum(~steps_ok & inner)
    is_run = (broken[offsets[1:]] - broken[offsets[:-1]] == 0) & (lens > 0) & (lens < 65536)
    out = []
    for length in np.unique(lens[mine]):
        sel = mine & (lens == length)
        out.append((int(length), int(sel.sum()), int((sel & is_run).sum())))
    return out


def build_image(db, *, shard_index: int = 0, shard_count: int = 1, free_bytes: int = 288 << 30, discard=False, **kw):
    """`epik_amd_placer_build_image`: the device image as three uint8 arrays (table, filter, postings);
    with `discard` the image is produced and dropped (returns the plan only).  Host only."""
    kw.setdefault("keys", getattr(db, "keys", None))
    kw.setdefault("holds_shard", getattr(db, "shard", None))
    desc, keep = make_desc(db.offsets, db.values, states=db.states, kmer_size=db.kmer_size,
                           num_branches=db.num_branches, threshold=db.threshold, log_threshold=db.log_threshold, **kw)
    lib = capi.load()
    p = capi.Plan()
    capi.check(lib.epik_amd_placer_plan(ctypes.byref(desc), shard_index, shard_count, int(free_bytes), ctypes.byref(p)))
    parts = [None, None, None] if discard else [np.zeros(int(n), dtype=np.uint8)
                                                for n in (p.table_bytes, p.filter_bytes, p.posting_bytes)]
    ptr = [None if a is None or a.size == 0 else a.ctypes.data for a in parts]
    capi.check(lib.epik_amd_placer_build_image(ctypes.byref(desc), shard_index, shard_count, int(free_bytes), *ptr))
    del keep
    return (p, *parts)


class Placer:
    """MI355X placer.  `offsets`/`values` are the CSR database (host arrays),
    `branch_length`/`subtree_*` the per-post-order-id tree data (may be None when
    only raw rows are wanted)."""

    def __init__(self, offsets: np.ndarray, values: np.ndarray, *, states: str, kmer_size: int,
                 num_branches: int, threshold, log_threshold=None, keep_at_most: int = 7,
                 keep_factor: float = 0.01, device: int = 0, branch_length=None,
                 subtree_num_nodes=None, subtree_total_length=None, char_class=None,
                 shard_index: int = 0, shard_count: int = 1, keys=None, sparse: bool = False, holds_shard=None):
        lib = capi.load()
        sigma = alphabet.alphabet_size(states)
        self.states = states
        self.kmer_size = int(kmer_size)
        self.num_branches = int(num_branches)
        self.keep_at_most = int(keep_at_most)
        self.keep_factor = float(keep_factor)
        self.device = int(device)
        desc, keepalive = make_desc(
            offsets, values, states=states, kmer_size=kmer_size, num_branches=num_branches, threshold=threshold,
            log_threshold=log_threshold, keep_at_most=keep_at_most, keep_factor=keep_factor, device=device,
            char_class=char_class, keys=keys, sparse=sparse, holds_shard=holds_shard)
        handle = ctypes.c_void_p()
        # shard_count > 1: this placer keeps the posting lists of the codes with
        # code % shard_count == shard_index (k-mer-space shard, `epik_amd_placer_create_sharded`)
        self.shard_index, self.shard_count = int(shard_index), int(shard_count)
        capi.check(lib.epik_amd_placer_create_sharded(ctypes.byref(desc), self.shard_index, self.shard_count,
                                                      ctypes.byref(handle)))
        del keepalive  # create() has streamed the database to the device and keeps no host copy
        self._lib = lib
        self._handle = handle
        if branch_length is not None:
            if len(branch_length) != self.num_branches:
                # place.cpp:104-108: "Could not find node by post-order id"
                raise RuntimeError(
                    f"Could not find node by post-order id: {min(len(branch_length), self.num_branches)}")
            self.distal, self.pendant = pendant_lengths(branch_length, subtree_num_nodes,
                                                        subtree_total_length)
        else:
            self.distal = self.pendant = None

    @classmethod
    def from_synth(cls, db, tree=None, **kw):
        extra = {}
        if tree is not None:
            extra = dict(branch_length=tree.branch_length, subtree_num_nodes=tree.subtree_num_nodes,
                         subtree_total_length=tree.subtree_total_length)
        kw.setdefault("keys", getattr(db, "keys", None))
        kw.setdefault("holds_shard", getattr(db, "shard", None))   # (synth.make_db(shard=...): one shard's lists only)
        return cls(db.offsets, db.values, states=db.states, kmer_size=db.kmer_size,
                   num_branches=db.num_branches, threshold=db.threshold,
                   log_threshold=db.log_threshold, **extra, **kw)

    # -- lifetime -----------------------------------------------------------------
    def close(self):
        if getattr(self, "_handle", None):
            self._lib.epik_amd_placer_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- the boundary, raw ----------------------------------------------------------
    def place_packed(self, seqs: np.ndarray, seq_offsets: np.ndarray):
        """Host buffers in, host buffers out (`epik_amd_placer_place`).  Returns
        (rows[n, keep] PLACEMENT, n_rows[n] uint32, kmer_counts[n, keep] uint32)."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        seq_offsets = np.ascontiguousarray(seq_offsets, dtype=np.uint64)
        n = int(seq_offsets.shape[0] - 1)
        rows = np.zeros((n, self.keep_at_most), dtype=capi.PLACEMENT)
        n_rows = np.zeros(n, dtype=np.uint32)
        counts = np.zeros((n, self.keep_at_most), dtype=np.uint32)
        capi.check(self._lib.epik_amd_placer_place(
            self._handle, seqs.ctypes.data, seq_offsets.ctypes.data, n, rows.ctypes.data,
            n_rows.ctypes.data, counts.ctypes.data))
        return rows, n_rows, counts

    def place_device(self, d_seqs: int, d_seq_offsets: int, n: int, d_rows: int, d_n_rows: int,
                     d_kmer_counts: int = 0, stream: int = 0) -> None:
        """Device pointers in and out, asynchronous on `stream` (`epik_amd_placer_place_device`)."""
        capi.check(self._lib.epik_amd_placer_place_device(
            self._handle, d_seqs, d_seq_offsets, int(n), d_rows, d_n_rows, d_kmer_counts or None,
            stream or None))

    def accumulate_device(self, d_seqs: int, d_seq_offsets: int, n: int, d_scores: int, d_counts: int,
                          stream: int = 0, d_amb_slot: int = 0, d_amb_order: int = 0, d_amb_avg: int = 0) -> None:
        """First half of a k-mer-space-sharded placement: raw float32 score sums and uint16 k-mer counts
        of this shard's lists, [n][num_branches] each, and -- for the reads with a slot in d_amb_slot --
        the records of their ambiguous k-mers (`epik_amd_placer_accumulate_device`)."""
        capi.check(self._lib.epik_amd_placer_accumulate_device(
            self._handle, d_seqs, d_seq_offsets, int(n), d_scores, d_counts, d_amb_slot or None,
            d_amb_order or None, d_amb_avg or None, stream or None))

    def finish_device(self, d_seq_offsets: int, n: int, d_scores: int, d_counts: int, d_rows: int,
                      d_n_rows: int, d_kmer_counts: int = 0, stream: int = 0, d_amb_slot: int = 0,
                      d_amb_avg: int = 0) -> None:
        """Second half: correction, top-k and like-weight-ratio on the sums added over the shards
        (`epik_amd_placer_finish_device`)."""
        capi.check(self._lib.epik_amd_placer_finish_device(
            self._handle, d_seq_offsets, int(n), d_scores, d_counts, d_amb_slot or None, d_amb_avg or None,
            d_rows, d_n_rows, d_kmer_counts or None, stream or None))

    def partial_info(self) -> dict:
        """Geometry of the partial lists of this handle (`epik_amd_placer_partial_info`)."""
        info = capi.PartialInfo()
        capi.check(self._lib.epik_amd_placer_partial_info(self._handle, ctypes.byref(info)))
        return {"lists": bool(info.lists), "slices": int(info.slices), "slice_rows": int(info.slice_rows),
                "entry_bytes": int(info.entry_bytes), "num_branches": int(info.num_branches),
                "postings_per_kmer": float(info.postings_per_kmer)}

    def accumulate_lists_device(self, d_seqs: int, d_seq_offsets: int, n: int, n_parts: int, d_entries: int,
                                entries_cap: int, d_index: int, d_part_entries: int, stream: int = 0,
                                d_amb_slot: int = 0, d_amb_order: int = 0, d_amb_avg: int = 0) -> None:
        """First half of a k-mer-space-sharded placement with partial LISTS: per read and slice of the branch
        range only the rows this shard's lists touched (`epik_amd_placer_accumulate_lists_device`)."""
        capi.check(self._lib.epik_amd_placer_accumulate_lists_device(
            self._handle, d_seqs, d_seq_offsets, int(n), int(n_parts), d_entries or None, int(entries_cap), d_index,
            d_part_entries, d_amb_slot or None, d_amb_order or None, d_amb_avg or None, stream or None))

    def finish_lists_device(self, d_seq_offsets: int, n: int, d_entries, d_index, d_rows: int, d_n_rows: int,
                            d_kmer_counts: int = 0, stream: int = 0, d_amb_slot: int = 0, d_amb_avg: int = 0) -> None:
        """Second half: the shards' lists of the n reads (d_entries[g], d_index[g]: device addresses, one per
        shard, in shard order) added in that order, then correction, top-k and like-weight-ratio
        (`epik_amd_placer_finish_lists_device`)."""
        g = len(d_entries)
        assert len(d_index) == g
        entries = (ctypes.c_void_p * g)(*[int(x) or None for x in d_entries])
        index = (ctypes.c_void_p * g)(*[int(x) or None for x in d_index])
        capi.check(self._lib.epik_amd_placer_finish_lists_device(
            self._handle, d_seq_offsets, int(n), g, entries, index, d_amb_slot or None, d_amb_avg or None,
            d_rows, d_n_rows, d_kmer_counts or None, stream or None))

    @staticmethod
    def place_sharded(placers, seqs: np.ndarray, seq_offsets: np.ndarray):
        """`epik_amd_placer_place_sharded`: host reads placed on `placers`, handle g holding shard g of
        len(placers) of one database (any devices).  Returns like `place_packed`."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        seq_offsets = np.ascontiguousarray(seq_offsets, dtype=np.uint64)
        n = int(seq_offsets.shape[0] - 1)
        keep = placers[0].keep_at_most
        rows = np.zeros((n, keep), dtype=capi.PLACEMENT)
        n_rows = np.zeros(n, dtype=np.uint32)
        counts = np.zeros((n, keep), dtype=np.uint32)
        handles = (ctypes.c_void_p * len(placers))(*[p._handle for p in placers])
        capi.check(placers[0]._lib.epik_amd_placer_place_sharded(
            handles, len(placers), seqs.ctypes.data, seq_offsets.ctypes.data, n, rows.ctypes.data, n_rows.ctypes.data,
            counts.ctypes.data))
        return rows, n_rows, counts

    def release_scratch(self) -> None:
        """Frees what the handle's launches have grown and kept (`epik_amd_placer_release_scratch`)."""
        capi.check(self._lib.epik_amd_placer_release_scratch(self._handle))

    def last_path(self) -> int:
        """Which kernels the last launch ran: capi.PATH_WAVE / PATH_TEAM_ONE_KERNEL / PATH_TEAM_STREAMED."""
        out = ctypes.c_uint32(0)
        capi.check(self._lib.epik_amd_placer_last_path(self._handle, ctypes.byref(out)))
        return int(out.value)

    def stream_build(self) -> dict:
        """Which build of the streaming kernel a large-tree placer launches with its current count width, and how many
        touched quads an item may have to take the touched-quad epilogue (`epik_amd_placer_stream_build`)."""
        wide, quads = ctypes.c_uint32(0), ctypes.c_uint32(0)
        capi.check(self._lib.epik_amd_placer_stream_build(self._handle, ctypes.byref(wide), ctypes.byref(quads)))
        return {"wide": bool(wide.value), "sparse_quads": int(quads.value)}

    def choose_counts(self, longest_read: int) -> None:
        """Width of the per-branch counts for the device entry points, chosen as `place_packed`
        chooses it from the batch (`epik_amd_placer_choose_counts`)."""
        capi.check(self._lib.epik_amd_placer_choose_counts(self._handle, int(longest_read)))

    def algorithmic_bytes(self, d_seqs: int, d_seq_offsets: int, n: int, d_n_rows: int = 0,
                          stream: int = 0) -> int:
        out = ctypes.c_uint64(0)
        capi.check(self._lib.epik_amd_placer_algorithmic_bytes(
            self._handle, d_seqs, d_seq_offsets, int(n), d_n_rows or None, stream or None,
            ctypes.byref(out)))
        return int(out.value)

    def launch_info(self) -> dict:
        w, b, l = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        capi.check(self._lib.epik_amd_placer_launch_info(self._handle, ctypes.byref(w),
                                                         ctypes.byref(b), ctypes.byref(l)))
        return {"waves_per_block": w.value, "blocks": b.value, "lds_bytes_per_block": l.value}

    def set_timing(self, enabled: bool) -> None:
        capi.check(self._lib.epik_amd_placer_set_timing(self._handle, int(bool(enabled))))

    def last_kernel_ms(self) -> float:
        ms = ctypes.c_float(-1.0)
        capi.check(self._lib.epik_amd_placer_last_kernel_ms(self._handle, ctypes.byref(ms)))
        return float(ms.value)

    # -- epik::placer::place ---------------------------------------------------------
    def place(self, seq_records: Iterable[Tuple[str, str]], num_threads: int = 1) -> PlacedCollection:
        """`seq_records` = (header, sequence) pairs (i2l::seq_record).  `num_threads`
        is accepted for signature parity and ignored, as the parallelism is the GPU's."""
        del num_threads
        sequence_map: dict = {}
        for header, sequence in seq_records:          # place.cpp:73-81
            sequence_map.setdefault(sequence, []).append(header)
        unique = list(sequence_map.keys())            # place.cpp:52-63
        bufs = [s.encode() for s in unique]
        offsets = np.zeros(len(bufs) + 1, dtype=np.uint64)
        if bufs:
            offsets[1:] = np.cumsum([len(b) for b in bufs], dtype=np.uint64)
        data = np.frombuffer(b"".join(bufs), dtype=np.uint8) if bufs else np.zeros(0, np.uint8)
        rows, n_rows, counts = self.place_packed(data, offsets)
        if len(n_rows) and int(n_rows.max()) > self.keep_at_most:
            # (never from place_packed, which widens the counts by itself: a row count, not the
            # EPIK_AMD_ROWS_COUNTS_TOO_NARROW mark of the device entry points)
            raise RuntimeError(f"a read came back with {int(n_rows.max())} rows (keep_at_most {self.keep_at_most})")
        placed = []
        for i, seq in enumerate(unique):
            pl = []
            for r in range(int(n_rows[i])):
                b = int(rows[i, r]["branch"])
                in_tree = self.distal is not None and b < self.num_branches
                pl.append(Placement(
                    branch_id=b, score=float(rows[i, r]["score"]),
                    weight_ratio=float(rows[i, r]["lwr"]), count=int(counts[i, r]),
                    # rows fabricated for a read without hits carry 0.0 lengths (place.cpp:150)
                    distal_length=float(self.distal[b]) if in_tree and counts[i, r] else 0.0,
                    pendant_length=float(self.pendant[b]) if in_tree and counts[i, r] else 0.0))
            placed.append(PlacedSequence(sequence=seq, placements=pl))
        return PlacedCollection(sequence_map=sequence_map, placed_seqs=placed)
