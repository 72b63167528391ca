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
    if rank == 0:
        ref = Oracle.from_synth(db).place(data, offs, num_threads=1)
        assert np.array_equal(got[1], ref[1])
        valid = np.arange(ref[0].shape[1])[None, :] < ref[1][:, None]
        same = got[0]["branch"][valid] == ref[0]["branch"][valid]
        assert same.mean() > 0.995, same.mean()
        np.testing.assert_allclose(got[0]["score"][valid], ref[0]["score"][valid], rtol=2e-6)
        assert np.abs(got[0]["lwr"][valid] - ref[0]["lwr"][valid]).max() <= 1e-5
        assert np.array_equal(got[2][valid][same], ref[2][valid][same])
        print(f"kmer-shard ok: world={world} reads={n} rows={int(valid.sum())}", flush=True)
    else:
        assert got is None
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
