#!/usr/bin/env python3
"""Race hunt for the team kernel (developer tool): a million reads -- a third of them with ambiguous or
invalid characters, lengths from below k to several tile groups -- placed by the team kernel and by the
one-wavefront kernel on the same device; every row must agree bit for bit (like-weight-ratios to 2e-6: the
slices add up sum_scores in another order),
run after run.  Also the accumulate + finish halves against the one-pass placement.

    python tools/stress_team.py [leaves] [reads] [repeats]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    torch.cuda.is_available()
    from epik_amd import alphabet, dist as edist, synth
    from epik_amd.placer import Placer
    leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    tree = synth.make_tree(leaves, seed=42)
    db = synth.make_db(tree.num_nodes, kmer_size=9, seed=43, lognormal=(3.3, 1.6))
    rng = np.random.default_rng(7)
    lengths = rng.integers(5, 420, size=n)
    offs = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    data = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(offs[-1]), dtype=np.uint8)]
    dirty = rng.choice(int(offs[-1]), size=int(offs[-1]) // 400, replace=False)
    data[dirty] = np.frombuffer(b"NRYKM-*", dtype=np.uint8)[rng.integers(0, 7, size=len(dirty))]

    def same_rows(got, want, what):
        """Branches, float32 scores and k-mer counts bit for bit; like-weight ratios to 2e-6: which epilogue a slice
        takes (dense, or over its touched quads) decides the order in which its share of sum_scores is added up, and
        the halves of a sharded placement finish in the streaming kernel the reads that a one-pass placement leaves
        to team_place_kernel (ambiguous k-mers)."""
        rows, n_rows, counts = got
        assert np.array_equal(n_rows, want[1]), (what, "row counts")
        valid = np.arange(rows.shape[1])[None, :] < n_rows[:, None]
        for field in ("branch", "score"):
            a, b = rows[field][valid], want[0][field][valid]
            assert a.tobytes() == b.tobytes(), (what, field, int((a.view(np.uint32) != b.view(np.uint32)).sum()))
        assert np.array_equal(counts[valid], want[2][valid]), (what, "counts")
        assert np.abs(rows["lwr"][valid] - want[0]["lwr"][valid]).max() <= 2e-6, (what, "lwr")

    def place(kernel):
        os.environ["EPIK_AMD_KERNEL"] = kernel
        with Placer.from_synth(db) as pl:
            return [pl.place_packed(data, offs) for _ in range(repeats)], pl.launch_info()

    ref, info = place("wave")
    print("wave  :", info, flush=True)
    # (team4: the wide streaming kernel with its epilogue over the touched quads at this tree size; team2 / team8 /
    # team4x2: the lean one; EPIK_AMD_TEAM_SPARSE=always: the touched-quad epilogue on every item its list holds)
    for kernel in ("team4", "team2", "team8", "team4x2", "team4+sparse"):
        if kernel.endswith("+sparse"):
            os.environ["EPIK_AMD_TEAM_SPARSE"] = "always"
            kernel = kernel[:-len("+sparse")]
        else:
            os.environ.pop("EPIK_AMD_TEAM_SPARSE", None)
        got, info = place(kernel)
        print(f"{kernel:6s}:", info, flush=True)
        for run, (rows, n_rows, counts) in enumerate(got):
            assert np.array_equal(n_rows, ref[0][1]), (kernel, run, "row counts")
            valid = np.arange(rows.shape[1])[None, :] < n_rows[:, None]
            for field in ("branch", "score"):
                a, b = rows[field][valid], ref[0][0][field][valid]
                assert a.tobytes() == b.tobytes(), (kernel, run, field, int((a.view(np.uint32) != b.view(np.uint32)).sum()))
            assert np.array_equal(counts[valid], ref[0][2][valid]), (kernel, run, "counts")
            assert np.abs(rows["lwr"][valid] - ref[0][0]["lwr"][valid]).max() <= 2e-6, (kernel, run, "lwr")
    os.environ.pop("EPIK_AMD_TEAM_SPARSE", None)
    print(f"one-pass: {n} reads x {repeats} runs x 5 team geometries agree with the one-wavefront kernel", flush=True)

    m = min(n, 200_000)
    sub_offs = offs[:m + 1]
    sub = data[:int(sub_offs[-1])]
    dev = torch.device("cuda", 0)
    os.environ["EPIK_AMD_KERNEL"] = "team4"
    with Placer.from_synth(db) as pl:
        one_pass = pl.place_packed(sub, sub_offs)
        slot, per = edist.amb_slots(sub, sub_offs, alphabet.char_class_table(db.states), 1)
        for run in range(repeats):
            accumulate, finish = edist.kmer_sharded_gpu_fns(pl, sub, sub_offs, dev)
            got = edist.place_kmer_sharded(accumulate, finish, m, None, amb_slot=slot, amb_per_owner=per)
            same_rows(got, one_pass, ("accumulate+finish", run))
    print(f"accumulate + finish: {m} reads x {repeats} runs agree with the one-pass placement", flush=True)

    # the same halves with partial lists (round 3), through the pipelined driver of epik_amd.dist in four batches, and
    # the whole sharded placement inside the library (one shard: bit for bit the one-pass rows)
    cls = alphabet.char_class_table(db.states)
    with Placer.from_synth(db) as pl:
        engine = edist.ListsGpuEngine(pl, dev)
        cuts = [0, m // 7, m // 2, m - 1000, m]
        for run in range(repeats):
            batches = [(sub[int(sub_offs[a]):int(sub_offs[b])], (sub_offs[a:b + 1] - sub_offs[a]).astype(np.uint64))
                       for a, b in zip(cuts, cuts[1:])]
            got = list(edist.place_kmer_sharded_lists(engine, batches, None, char_class=cls))
            same_rows(tuple(np.concatenate([g[i] for g in got]) for i in range(3)), one_pass, ("accumulate_lists + finish_lists", run))
            os.environ["EPIK_AMD_SHARD_HALVES"] = "1"   # (one handle would take the one-pass placement by itself)
            same_rows(Placer.place_sharded([pl], sub, sub_offs), one_pass, ("place_sharded, one shard", run))
            os.environ.pop("EPIK_AMD_SHARD_HALVES")
    print(f"partial lists: {m} reads x {repeats} runs (pipelined halves and place_sharded) agree with the one-pass placement", flush=True)
    # three shards on the one device: the library's own exchange against the halves driven from here, bit for bit
    os.environ["EPIK_AMD_SHARD_CHUNK"] = "20000"
    placers = [Placer.from_synth(db, shard_index=g, shard_count=3) for g in range(3)]
    try:
        first = Placer.place_sharded(placers, sub, sub_offs)
        for run in range(repeats):
            again = Placer.place_sharded(placers, sub, sub_offs)
            for a, b in zip(first, again):
                assert a.tobytes() == b.tobytes(), ("place_sharded, three shards, run to run", run)
        assert np.array_equal(first[1], one_pass[1])
        valid = np.arange(first[0].shape[1])[None, :] < first[1][:, None]
        np.testing.assert_allclose(first[0]["score"][valid], one_pass[0]["score"][valid], rtol=2e-6)
        assert np.abs(first[0]["lwr"][valid] - one_pass[0]["lwr"][valid]).max() <= 1e-5
    finally:
        for p in placers:
            p.close()
    print(f"place_sharded, 3 shards: {m} reads x {repeats + 1} runs identical, within rounding of the one-pass rows", flush=True)


if __name__ == "__main__":
    main()
