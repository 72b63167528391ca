"""CPU tests of the oracle itself (no GPU): the C restatement against the
independently written numpy restatement and against hand-derived values."""
import json
import math
import os

import numpy as np
import pytest

from conftest import mixed_reads
from epik_amd import alphabet, synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
f32 = np.float32


def test_threshold_formula():
    # nucl k=10 omega=1.5 -> 0.375^10 = 5.4994e-5, log10 = -4.2597
    thr = alphabet.score_threshold(1.5, 10, 4)
    assert abs(float(thr) - 0.375 ** 10) < 1e-11
    assert abs(float(alphabet.log_threshold(thr)) - (-4.2597)) < 1e-4


def test_c_oracle_matches_numpy_restatement(oracle_lib, small_case):
    from oracle.epik_oracle_np import RefShapedPlacer, dict_db_from_csr
    tree, db = small_case
    orc = oracle_lib.Oracle.from_synth(db)
    rng = np.random.default_rng(0)
    reads = mixed_reads(rng, 300, db.kmer_size)
    reads += ["ACG", "ACGT", "NNNNNNNN", "ACGTACGTNACGT", "-" * 10, "acgtacgtacgu"]
    data, offs = synth.pack_reads(reads)
    rows, n_rows, counts = orc.place(data, offs)
    ref = RefShapedPlacer(dict_db_from_csr(db.offsets, db.values), kmer_size=db.kmer_size,
                          alphabet_size=4, num_branches=tree.num_nodes, threshold=db.threshold,
                          log_threshold=db.log_threshold,
                          char_class=alphabet.char_class_table("nucl"))
    for i, r in enumerate(reads):
        out = ref.place(r.encode())
        if out is None:
            assert n_rows[i] == 0
            continue
        assert len(out) == n_rows[i], (i, r)
        for j, (b, s, lwr, c) in enumerate(out):
            x = rows[i, j]
            assert x["branch"] == b
            assert x["score"].view(np.uint32) == f32(s).view(np.uint32)
            assert x["lwr"] == lwr        # same pow, same summation order: identical doubles
            assert counts[i, j] == c


def test_c_oracle_threads_agree(oracle_lib, small_case):
    tree, db = small_case
    orc = oracle_lib.Oracle.from_synth(db)
    rng = np.random.default_rng(3)
    data, offs = synth.pack_reads(mixed_reads(rng, 500, db.kmer_size))
    a = orc.place(data, offs, num_threads=1)
    b = orc.place(data, offs, num_threads=4)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def _micro_db():
    """N=5, k=3 nucl micro database; every value below is derived by hand in
    test_hand_derived_micro_case from the formulas of place.cpp."""
    sigma, k, n = 4, 3, 5
    lists = {
        "ACG": [(0, -0.5), (2, -1.0)],
        "CGT": [(2, -0.25), (3, -2.0)],
        "GTA": [(4, -0.125)],
    }
    code = {c: i for i, c in enumerate("ACGT")}
    num_keys = sigma ** k
    lens = np.zeros(num_keys, dtype=np.int64)
    for kmer, lst in lists.items():
        key = 0
        for ch in kmer:
            key = key * sigma + code[ch]
        lens[key] = len(lst)
    offsets = np.zeros(num_keys + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    values = np.zeros(int(offsets[-1]), dtype=synth.PKDB_VALUE)
    for kmer, lst in lists.items():
        key = 0
        for ch in kmer:
            key = key * sigma + code[ch]
        b = int(offsets[key])
        for j, (br, sc) in enumerate(lst):
            values[b + j] = (br, sc)
    return synth.SynthDB(states="nucl", kmer_size=k, omega=1.5, num_branches=n,
                         offsets=offsets, values=values)


def test_hand_derived_micro_case(oracle_lib):
    """Read ACGTA over the micro DB: k-mers ACG, CGT, GTA (all found).
    Derivation, float32 unless noted (place.cpp:349-371, 418-422, 164-184, 241-267):
      threshold t = (1.5/4)^3, L = log10f(t), n_kmers = 3, k = 3
      branch 0: sum=-0.5   c=1 -> (-0.5   + 2*L)/3
      branch 2: sum=-1.25  c=2 -> (-1.25  + 1*L)/3
      branch 3: sum=-2.0   c=1 -> (-2.0   + 2*L)/3
      branch 4: sum=-0.125 c=1 -> (-0.125 + 2*L)/3
      S = (5-4) * 10^(3*L/3) + sum_b 10^score_b        (double)
      rows sorted by score desc; lwr = 10^score / S; keep lwr >= 0.01 * best."""
    db = _micro_db()
    L = db.log_threshold
    k = f32(3)

    def corrected(total, count):
        return f32(f32(f32(total) + f32(f32(3 - count) * L)) / k)

    expect = {0: corrected(-0.5, 1), 2: corrected(-1.25, 2), 3: corrected(-2.0, 1),
              4: corrected(-0.125, 1)}
    S = (5.0 - 4.0) * math.pow(10.0, float(f32(f32(f32(3) * L) / k)))
    for b in (0, 2, 3, 4):       # edge insertion order: ACG -> 0, 2; CGT -> 3; GTA -> 4
        S += math.pow(10.0, float(expect[b]))
    order = sorted(expect, key=lambda b: (-float(expect[b]), b))
    lwr = {b: math.pow(10.0, float(expect[b])) / S for b in order}
    kept = [b for b in order if lwr[b] >= lwr[order[0]] * 0.01]

    orc = oracle_lib.Oracle.from_synth(db)
    data, offs = synth.pack_reads(["ACGTA"])
    rows, n_rows, counts = orc.place(data, offs)
    assert n_rows[0] == len(kept)
    for j, b in enumerate(kept):
        assert rows[0, j]["branch"] == b
        assert rows[0, j]["score"].view(np.uint32) == expect[b].view(np.uint32)
        assert rows[0, j]["lwr"] == pytest.approx(lwr[b], rel=1e-15)
    assert list(counts[0, :len(kept)]) == [{0: 1, 2: 2, 3: 1, 4: 1}[b] for b in kept]
    # sanity of the hand numbers: branch 2 (two hits) beats branch 4; LWRs sum to < 1
    assert kept[0] == 2 and kept[1] == 4 and sum(lwr.values()) < 1.0


def test_zero_hit_read_fabricates_first_branches(oracle_lib):
    """No k-mer found -> keep_at_most rows for branches 0..6 at the threshold score,
    LWR = 1/N each (place.cpp:141-152, 174-175)."""
    db = _micro_db()
    orc = oracle_lib.Oracle.from_synth(db)
    data, offs = synth.pack_reads(["TTTTTT"])
    rows, n_rows, counts = orc.place(data, offs)
    assert n_rows[0] == 7
    assert list(rows[0]["branch"]) == list(range(7))
    thr_score = f32(f32(db.log_threshold * f32(4)) / f32(3))
    assert all(rows[0]["score"].view(np.uint32) == thr_score.view(np.uint32))
    assert np.allclose(rows[0]["lwr"], 1.0 / 5.0, rtol=1e-15)
    assert not counts[0].any()


def test_ambiguous_kmer_quirks(oracle_lib):
    """ACGNA: windows ACG (exact), CGN (ambiguous -> CGA,CGC,CGG,CGT), GNA
    (ambiguous -> GAA,GCA,GGA,GTA).  Found keys: CGT -> branches 2,3; GTA -> 4.
    place.cpp:395-402: avg = (10^score + (k-1)*threshold)/k added to a log sum."""
    db = _micro_db()
    t, L, k = db.threshold, db.log_threshold, f32(3)

    def avg(score):
        p = f32(math.pow(10.0, float(f32(score))))
        return f32(f32(p + f32(f32(2) * t)) / k)

    def corrected(total, count):
        return f32(f32(f32(total) + f32(f32(3 - count) * L)) / k)

    expect = {0: corrected(f32(-0.5), 1),
              2: corrected(f32(f32(-1.0) + avg(-0.25)), 2),
              3: corrected(avg(-2.0), 1),
              4: corrected(avg(-0.125), 1)}
    orc = oracle_lib.Oracle.from_synth(db, keep_factor=0.0)
    data, offs = synth.pack_reads(["ACGNA"])
    rows, n_rows, counts = orc.place(data, offs)
    got = {int(r["branch"]): r["score"] for r in rows[0, :n_rows[0]]}
    assert set(got) == set(expect)
    for b, s in expect.items():
        assert got[b].view(np.uint32) == s.view(np.uint32), b


def test_short_read_reports_no_placement(oracle_lib):
    db = _micro_db()
    orc = oracle_lib.Oracle.from_synth(db)
    data, offs = synth.pack_reads(["AC", "", "ACG"])
    rows, n_rows, _ = orc.place(data, offs)
    assert list(n_rows[:2]) == [0, 0] and n_rows[2] > 0


def test_golden_fixture_regression(oracle_lib):
    """tests/golden/synth_k6.json was written by tests/golden/make_golden.py from
    the C oracle after it agreed with the numpy restatement; it freezes those
    outputs so that later edits to either cannot drift silently."""
    path = os.path.join(GOLDEN, "synth_k6.json")
    with open(path) as fh:
        g = json.load(fh)
    tree = synth.make_tree(g["n_leaves"], seed=g["tree_seed"])
    db = synth.make_db(tree.num_nodes, kmer_size=g["kmer_size"], p_present=g["p_present"],
                       seed=g["db_seed"], lognormal=tuple(g["lognormal"]))
    orc = oracle_lib.Oracle.from_synth(db)
    data, offs = synth.pack_reads(g["reads"])
    rows, n_rows, counts = orc.place(data, offs)
    assert list(map(int, n_rows)) == g["n_rows"]
    for i, exp in enumerate(g["rows"]):
        for j, (b, score_bits, lwr, c) in enumerate(exp):
            assert int(rows[i, j]["branch"]) == b
            assert int(rows[i, j]["score"].view(np.uint32)) == score_bits
            assert float(rows[i, j]["lwr"]) == pytest.approx(lwr, rel=1e-14, abs=1e-300)
            assert int(counts[i, j]) == c


def test_hash_map_lookup_and_batched_dedup_change_nothing(oracle_lib, small_case):
    """The two CPU-baseline variants of bench.py: phylo_kmer_db::search through a
    node-chained hash map, and placer::place as the driver runs it -- batches of `batch_size` reads,
    each de-duplicated by content (place.cpp:73-81, 207-212) -- give the rows of the plain loop."""
    _, db = small_case
    rng = np.random.default_rng(3)
    reads = mixed_reads(rng, 3000, db.kmer_size, max_len=120)
    reads += reads[:500] + ["", "AC", reads[7]]          # duplicates inside and across batches
    data, offs = synth.pack_reads(reads)
    orc = oracle_lib.Oracle.from_synth(db)
    plain = orc.place(data, offs, num_threads=2)
    for use_hash in (False, True):
        orc.use_hash_map(use_hash)
        for got in (orc.place(data, offs, num_threads=2), orc.place_batched(data, offs, batch_size=700, num_threads=3)):
            for a, b in zip(plain, got):
                assert a.tobytes() == b.tobytes()


def test_hash_map_from_the_sparse_form_gives_the_same_placements(oracle_lib):
    """Oracle.from_sparse: search() through a hash map built from keys[present] + offsets[present + 1] -- no array per
    possible code -- against the direct index over the densified database (what tests at amino k = 7's stated size
    check the device against)."""
    from epik_amd import synth
    tree = synth.make_tree(50, seed=1)
    db = synth.make_sparse_db(tree.num_nodes, states="amino", kmer_size=4, p_present=0.05, seed=3, dense=False)
    dense = db.densified()
    data, offs = synth.reads_hitting(db, 300, 60, hit_rate=0.3, seed=5, dirty="BZXJ*")
    sparse_rows = oracle_lib.Oracle.from_sparse(db).place(data, offs)
    dense_rows = oracle_lib.Oracle.from_synth(dense).place(data, offs)
    assert (dense_rows[1] > 0).mean() > 0.9
    for a, b in zip(sparse_rows, dense_rows):
        assert a.tobytes() == b.tobytes()
