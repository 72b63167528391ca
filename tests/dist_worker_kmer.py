"""Worker of tests/test_dist_cpu.py::test_gloo_kmer_sharded...: the exchange step of the
k-mer-space shard (epik_amd.dist.place_kmer_sharded: all-to-all + sum in rank order) under
torch.distributed.run with gloo.  On CPU a numpy engine stands in for the two kernel halves
(test infrastructure, built on the oracle's restatement); with EPIK_AMD_DIST_GPU=1 the HIP
placer on device 0 is used, the partial vectors crossing in host memory."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from epik_amd import alphabet, dist as edist, synth  # noqa: E402
from oracle.epik_oracle_np import RefShapedPlacer, dict_db_from_csr  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

f32 = np.float32


def numpy_engine(db, data, offs, rank, world):
    """accumulate / finish over the pure-Python restatement, ambiguous k-mers included: a shard
    records, per branch, the first ambiguous key of ITS lists that reaches it (place.cpp:385-388)."""
    import torch
    n, N, k = len(offs) - 1, db.num_branches, db.kmer_size
    ref = RefShapedPlacer(dict_db_from_csr(db.offsets, db.values), kmer_size=k, alphabet_size=4, num_branches=N,
                          threshold=db.threshold, log_threshold=db.log_threshold,
                          char_class=alphabet.char_class_table("nucl"))

    def accumulate(n_rows_padded, amb_slot, amb_rows):
        scores = np.zeros((n_rows_padded, N), dtype=np.float32)
        counts = np.zeros((n_rows_padded, N), dtype=np.uint16)
        order = np.full((amb_rows, N), edist.AMB_NONE, dtype=np.uint32) if amb_rows else None
        avg = np.zeros((amb_rows, N), dtype=np.float32) if amb_rows else None
        for i in range(n):
            seq = bytes(data[int(offs[i]):int(offs[i + 1])])
            position = 0
            for window, keys in ref.to_kmers_positions(seq):
                if len(keys) == 1:
                    if keys[0] % world == rank:
                        for branch, score in ref.db.get(keys[0]) or ():
                            scores[i, branch] = f32(scores[i, branch] + f32(score))
                            counts[i, branch] += 1
                    continue
                slot = int(amb_slot[i])
                assert slot >= 0, "a read with an ambiguous k-mer must have a slot"
                for state, key in keys:      # ascending state order
                    if key % world != rank:
                        continue
                    for branch, score in ref.db.get(key) or ():
                        if order[slot, branch] == edist.AMB_NONE:
                            prob = f32(10.0 ** float(f32(score)))
                            order[slot, branch] = window * 4 + state
                            avg[slot, branch] = f32(f32(prob + f32(f32(k - 1) * f32(db.threshold))) / f32(k))
        to = torch.from_numpy
        return (to(scores), to(counts.view(np.int16)), None if order is None else to(order.view(np.int32)),
                None if avg is None else to(avg))

    def finish(begin, end, scores, counts, amb_slot, amb_avg):
        scores, counts = scores.numpy().copy(), counts.numpy().view(np.uint16).astype(np.int64)
        m, keep = end - begin, ref.keep_at_most
        rows = np.zeros((m, keep), dtype=[("branch", np.uint32), ("score", np.float32), ("lwr", np.float64)])
        n_rows = np.zeros(m, dtype=np.uint32)
        kc = np.zeros((m, keep), dtype=np.uint32)
        for j in range(m):
            length = int(offs[begin + j + 1] - offs[begin + j])
            if length < k:
                continue
            if amb_avg is not None and amb_slot[j] >= 0:   # the one record per branch, after the exact scores
                a = amb_avg[int(amb_slot[j])].numpy()
                hit = a > 0
                scores[j, hit] = (scores[j, hit] + a[hit]).astype(np.float32)
                counts[j, hit] += 1
            nk = length - k + 1
            touched = np.nonzero(counts[j])[0]
            placements = [(int(b), f32(f32(scores[j, b] + f32(f32(nk - counts[j, b]) * ref.log_threshold)) / f32(k)),
                           int(counts[j, b])) for b in touched]
            ref.place_seq = lambda seq, _p=placements: _p    # the epilogue of RefShapedPlacer.place on these sums
            out = ref.place(b"A" * length)
            n_rows[j] = len(out)
            for r, (b, s, lwr, c) in enumerate(out):
                rows[j, r] = (b, s, lwr)
                kc[j, r] = c
        return rows, n_rows, kc

    return accumulate, finish


class NumpyListsEngine:
    """The engine protocol of epik_amd.dist.place_kmer_sharded_lists over the numpy halves above: the dense
    vectors cut into `slices` slices and compacted into partial lists exactly as the header describes them
    (8-byte entries {f32 sum, u32 row | count << 16}, an index {first, count} per (read, slice), the parts one
    after the other), and read back by adding the shards' lists in shard order."""
    compute_stream = comm_stream = None
    host_staging = True

    def __init__(self, db, rank, world, slices=3, cap_entries=None):
        self.db, self.rank, self.world = db, rank, world
        self.slices, self.entry_bytes = slices, 8
        self.slice_rows = -(-db.num_branches // slices)
        self.cap_entries = cap_entries      # a first capacity that is too small exercises the overflow round
        self.margin = 1.0
        self.accumulate_calls = 0

    def begin(self, seqs, offs):
        acc, fin = numpy_engine(self.db, seqs, offs, self.rank, self.world)
        return {"n": len(offs) - 1, "acc": acc, "fin": fin}

    def accumulate(self, batch, n_parts, amb_slot, amb_rows, min_entries=0):
        import torch
        self.accumulate_calls += 1
        n, S, N = batch["n"], self.slices, self.db.num_branches
        per = -(-n // n_parts) if n else 0
        scores, counts, order, avg = batch["acc"](n, amb_slot if amb_rows else None, amb_rows)
        scores, counts = scores.numpy(), counts.numpy().view(np.uint16)
        index = np.zeros((per * n_parts, S, 2), dtype=np.uint32)
        part_entries = np.zeros(n_parts, dtype=np.int64)
        chunks = []
        for r in range(n_parts):
            at = 0
            for i in range(r * per, min(n, (r + 1) * per)):
                for s in range(S):
                    lo, hi = s * self.slice_rows, min(N, (s + 1) * self.slice_rows)
                    rows = np.nonzero(counts[i, lo:hi])[0]
                    rows = rows[::-1]  # (any order inside a list)
                    e = np.zeros((len(rows), 2), dtype=np.uint32)
                    e[:, 0] = scores[i, lo + rows].view(np.uint32)
                    e[:, 1] = rows.astype(np.uint32) | (counts[i, lo + rows].astype(np.uint32) << 16)
                    index[i, s] = (at, len(rows))
                    chunks.append(e)
                    at += len(rows) + (i % 3)  # (a part may take more room than its lists fill: upper bounds)
                    chunks.append(np.full((i % 3, 2), 0xDEADBEEF, dtype=np.uint32))
            part_entries[r] = at
        entries = np.concatenate(chunks).reshape(-1) if chunks else np.zeros(0, np.uint32)
        cap = max(int(min_entries), self.cap_entries if self.cap_entries is not None else int(part_entries.sum()))
        if part_entries.sum() > cap:      # what the kernel does: the lists that find no room are marked
            index[:, :, 1] = 0xFFFFFFFF
            entries = entries[:cap * 2]
        return edist.Partials(torch.from_numpy(entries.view(np.uint8).copy()), torch.from_numpy(index.view(np.int32)),
                              torch.from_numpy(part_entries), order, avg, cap)

    def finish(self, batch, begin, end, entries, index, amb_slot, amb_avg):
        import torch
        m, N, S = end - begin, self.db.num_branches, self.slices
        scores = np.zeros((m, N), dtype=np.float32)
        counts = np.zeros((m, N), dtype=np.int64)
        for e, ix in zip(entries, index):    # shard order: the float32 sums of the dense rank-order sum
            e = e.numpy().view(np.uint32).reshape(-1, 2)
            ix = ix.numpy().view(np.uint32)
            for j in range(m):
                for s in range(S):
                    first, count = int(ix[j, s, 0]), int(ix[j, s, 1])
                    assert count != 0xFFFFFFFF
                    rows = (e[first:first + count, 1] & 0xFFFF).astype(np.int64) + s * self.slice_rows
                    scores[j, rows] = (scores[j, rows] + e[first:first + count, 0].view(np.float32)).astype(np.float32)
                    counts[j, rows] += e[first:first + count, 1] >> 16
        return batch["fin"](begin, end, torch.from_numpy(scores), torch.from_numpy(counts.astype(np.uint16).view(np.int16)),
                            amb_slot, amb_avg)


def check(got, ref, rank, world, n, what):
    if rank == 0:
        assert np.array_equal(got[1], ref[1])
        valid = np.arange(ref[0].shape[1])[None, :] < ref[1][:, None]
        same = got[0]["branch"][valid] == ref[0]["branch"][valid]
        assert same.mean() > 0.995, same.mean()
        np.testing.assert_allclose(got[0]["score"][valid], ref[0]["score"][valid], rtol=2e-6)
        assert np.abs(got[0]["lwr"][valid] - ref[0]["lwr"][valid]).max() <= 1e-5
        assert np.array_equal(got[2][valid][same], ref[2][valid][same])
        print(f"{what} ok: world={world} reads={n} rows={int(valid.sum())}", flush=True)
    else:
        assert got is None


def main():
    rank, _, world = edist.env_rank_world()
    dist = edist.init_process_group("gloo")
    tree = synth.make_tree(8, seed=1)
    db = synth.make_db(tree.num_nodes, kmer_size=4, p_present=0.7, seed=5, lognormal=(1.0, 1.0))
    data, offs = synth.make_reads(203, 37, seed=3)      # not divisible by the world size
    data = data.copy()
    rng = np.random.default_rng(11)                     # ambiguous and invalid characters in a third of the reads
    for i in rng.choice(203, size=70, replace=False):
        data[int(offs[i]) + rng.integers(0, 37, size=2)] = rng.choice(np.frombuffer(b"NRYN-", dtype=np.uint8), size=2)
    n = len(offs) - 1
    amb_slot, amb_per_owner = edist.amb_slots(data, offs, alphabet.char_class_table("nucl"), world)
    if os.environ.get("EPIK_AMD_DIST_GPU") == "1":
        import torch
        from epik_amd.placer import Placer
        placer = Placer.from_synth(db, device=0, shard_index=rank, shard_count=world)
        accumulate, finish = edist.kmer_sharded_gpu_fns(placer, data, offs, torch.device("cuda", 0), host_staging=True)
    else:
        accumulate, finish = numpy_engine(db, data, offs, rank, world)
    got = edist.place_kmer_sharded(accumulate, finish, n, dist, gather_to=0, amb_slot=amb_slot,
                                   amb_per_owner=amb_per_owner)
    dist.barrier()
    oracle = Oracle.from_synth(db)
    ref = oracle.place(data, offs, num_threads=1) if rank == 0 else None
    check(got, ref, rank, world, n, "kmer-shard")

    # ---- the same with partial lists, three batches through the pipelined exchange (the middle one without
    # any ambiguous character, the last one short), the first capacity too small: one overflow round
    cuts = [0, 90, 150, n]
    batches = []
    for b in range(3):
        lo, hi = cuts[b], cuts[b + 1]
        seqs = data[int(offs[lo]):int(offs[hi])].copy()
        if b == 1:
            seqs[~np.isin(seqs, np.frombuffer(b"ACGT", dtype=np.uint8))] = ord("A")
        batches.append((seqs, (offs[lo:hi + 1] - offs[lo]).astype(np.uint64)))
    if os.environ.get("EPIK_AMD_DIST_GPU") == "1":
        info = placer.partial_info()
        if not info["lists"]:
            print(f"kmer-shard lists skipped (dense partials on this kernel): world={world}", flush=True)
            dist.destroy_process_group()
            return
        engine = edist.ListsGpuEngine(placer, torch.device("cuda", 0), host_staging=True)
        engine.margin = 0.05  # (the first batch overflows)
    elif os.environ.get("EPIK_AMD_TEST_ONE_RANK_OVERFLOWS") == "1":
        # only the last rank's first capacity is too small (capacities follow a shard's own lists: one shard
        # overflowing alone is the usual case) -- every rank must still take the same collectives
        engine = NumpyListsEngine(db, rank, world, cap_entries=40 if rank == world - 1 else None)
    else:
        engine = NumpyListsEngine(db, rank, world, cap_entries=40)
    results = list(edist.place_kmer_sharded_lists(engine, batches, dist, char_class=alphabet.char_class_table("nucl")))
    dist.barrier()
    assert len(results) == 3
    if os.environ.get("EPIK_AMD_DIST_GPU") != "1":
        assert engine.accumulate_calls > 3, "the overflow round was not taken"   # (on every rank: they repeat together)
    for b, got_b in enumerate(results):
        ref_b = oracle.place(*batches[b], num_threads=1) if rank == 0 else None
        check(got_b, ref_b, rank, world, cuts[b + 1] - cuts[b], f"kmer-shard lists batch {b}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
