"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU
oracle on the same seeded inputs.  Bar (BASELINE.json north_star): branch ids in
identical order, float32 scores bit-identical, |delta like_weight_ratio| <= 1e-5
(observed: ~1e-7, from the float32 relative accumulation of score_sum)."""
import json
import os

import numpy as np
import pytest

from conftest import assert_rows_match, mixed_reads, select_kernel
from epik_amd import alphabet, synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(autouse=True, params=["paired", "filtered", "packed", "compact", "paired-runs", "filtered-runs", "packed-runs",
                                      "team4", "team8", "team4x3", "team2", "team2x3", "team2-classic", "team2-smallpool", "team2-sparse", "team4-sparse", "team4x2-sparse", "team4-dense", "team4-block2", "team2x2-block2", "team8-block2", "team4-block2wide",
                                      "team4-classic", "team4x3-classic", "team4-smallpool", "team8x2-smallpool",
                                      "paired-fewblocks", "team4-fewblocks", "team8x2-fewblocks", "team4-classic-fewblocks"])
def db_layout(request, monkeypatch):
    """Every parity test runs on every HBM layout of the database (line-aligned lists behind a
    direct-index table -- keyed by k-mer, or by the overlap of consecutive k-mers for DNA, or behind
    a presence filter keyed that way -- and the CSR used when that table would not fit) with one
    wavefront per read, and with the team kernels (a workgroup of 4 / 8 waves per read over the sliced
    database; team4x3: the branch range in three passes of four slices), whatever the size of the tree:
    as front kernel + streaming kernel, as the one-kernel placement (-classic), and with a descriptor
    pool too small for the batch (-smallpool: both, mixed)."""
    select_kernel(monkeypatch, request.param)
    return request.param


def test_the_sparse_variants_reach_the_touched_quad_epilogue(gpu_available, small_case, db_layout):
    """teamW-sparse / -dense run the WIDE build of the streaming kernel on the small trees of these tests
    (EPIK_AMD_STREAM_WIDE=1), -sparse with the slice epilogue over the touched quads open to every item of up to 256
    touched quads -- every item of a tree of a few hundred branches: the tie overflow and the tiny-score division of
    team_epilogue.hpp are then what test_many_ties_overflow_candidate_buffer and test_underflowing_scores_zero_lwr
    exercise under these variants -- and -dense with it closed."""
    assert gpu_available
    from epik_amd.placer import Placer
    tree, db = small_case
    with Placer.from_synth(db) as pl:
        for longest in (40, 300, 40_000):   # 8-, 16-, 32-bit counts
            pl.choose_counts(longest)
            build = pl.stream_build()
            if db_layout.endswith(("-sparse", "-block2wide")):
                assert build["wide"] and build["sparse_quads"] == (256 if longest < 40_000 else 0), (longest, build)
            elif db_layout.endswith("-dense"):
                assert build["wide"] and build["sparse_quads"] == 0
            elif db_layout.startswith("team") and "classic" not in db_layout:
                assert not build["wide"]   # a small tree by itself: the lean build
            else:
                assert build == {"wide": False, "sparse_quads": 0}


@pytest.fixture(scope="module")
def placer_cls(gpu_available):
    assert gpu_available, "pytest -m gpu needs a HIP device (no CPU fallback exists)"
    from epik_amd.placer import Placer
    return Placer


def _compare(placer_cls, oracle_lib, db, data, offs, **kw):
    orc = oracle_lib.Oracle.from_synth(db, **kw)
    ref = orc.place(data, offs, num_threads=0)
    with placer_cls.from_synth(db, **kw) as pl:
        got = pl.place_packed(data, offs)
    return assert_rows_match(*got, *ref)


def test_small_db_mixed_reads(placer_cls, oracle_lib, small_case):
    _, db = small_case
    rng = np.random.default_rng(0)
    reads = mixed_reads(rng, 2000, db.kmer_size, max_len=300)
    reads += ["ACG", "", "ACGT", "NNNNNNNN", "ACGTACGTNACGT", "-" * 10, "acgtacgtacgu", "T" * 50]
    data, offs = synth.pack_reads(reads)
    _compare(placer_cls, oracle_lib, db, data, offs)


@pytest.mark.parametrize("chunks", ["1", "7", "256"])
def test_host_buffer_pipeline_chunks(placer_cls, oracle_lib, small_case, chunks, monkeypatch):
    """`epik_amd_placer_place` copies in, computes and copies out in chunks on three streams;
    the rows must not depend on how the batch was cut (ragged reads, empty reads at chunk
    edges, more chunks than the 256 it caps at)."""
    _, db = small_case
    monkeypatch.setenv("EPIK_AMD_HOST_CHUNKS", chunks)
    rng = np.random.default_rng(5)
    reads = mixed_reads(rng, 3001, db.kmer_size, max_len=400)
    for i in range(0, len(reads), 429):
        reads[i] = ""
    data, offs = synth.pack_reads(reads)
    _compare(placer_cls, oracle_lib, db, data, offs)


def test_create_through_small_staging_buffers(placer_cls, oracle_lib, monkeypatch):
    """`create()` streams the image through two pinned staging buffers (16 MB each); with 4 KB ones a
    small database crosses hundreds of buffer switches, and every list longer than a buffer (here:
    of more than ~680 postings) takes the path of a single posting list of millions of branches."""
    monkeypatch.setenv("EPIK_AMD_STAGE_BYTES", "4096")
    tree = synth.make_tree(600, seed=3)
    db = synth.make_db(tree.num_nodes, kmer_size=5, p_present=0.8, seed=8, lognormal=(5.5, 1.5))
    assert int(np.diff(db.offsets).max()) > 800
    rng = np.random.default_rng(12)
    data, offs = synth.pack_reads(mixed_reads(rng, 1500, db.kmer_size, max_len=200))
    _compare(placer_cls, oracle_lib, db, data, offs)


def test_golden_fixture(placer_cls):
    """Committed vectors (tests/golden/make_golden.py)."""
    with open(os.path.join(GOLDEN, "synth_k6.json")) as fh:
        g = json.load(fh)
    tree = synth.make_tree(g["n_leaves"], seed=g["tree_seed"])
    db = synth.make_db(tree.num_nodes, kmer_size=g["kmer_size"], p_present=g["p_present"],
                       seed=g["db_seed"], lognormal=tuple(g["lognormal"]))
    data, offs = synth.pack_reads(g["reads"])
    with placer_cls.from_synth(db) as pl:
        rows, n_rows, counts = pl.place_packed(data, offs)
    assert list(map(int, n_rows)) == g["n_rows"]
    for i, exp in enumerate(g["rows"]):
        for j, (b, score_bits, lwr, c) in enumerate(exp):
            assert int(rows[i, j]["branch"]) == b
            assert int(rows[i, j]["score"].view(np.uint32)) == score_bits
            assert abs(float(rows[i, j]["lwr"]) - lwr) <= 1e-5
            assert int(counts[i, j]) == c


@pytest.mark.parametrize("wide", ["0", "1", "2"])
def test_config1_shape_k10(placer_cls, oracle_lib, wide, monkeypatch):
    """BASELINE configs[0]/[1] shape at a size the oracle finishes in seconds:
    nucl k=10, N=1303 (652 leaves, the D652 substitute), 150 bp reads.  The 16-bit-count
    kernels (default), the 32-bit ("wide") and the 8-bit ones must all be bit-exact."""
    monkeypatch.setenv("EPIK_AMD_WIDE_COUNTS", wide)
    tree = synth.make_tree(652, seed=42)
    db = synth.make_db(tree.num_nodes, kmer_size=10, seed=43)
    data, offs = synth.make_reads(20000, 150, seed=44)
    worst = _compare(placer_cls, oracle_lib, db, data, offs)
    # score_sum is accumulated relative to its largest term in float32 (v_exp_f32):
    # observed |delta LWR| ~1e-7; the bar is 1e-5
    assert worst < 2e-6


def test_scattered_branches_and_long_reads(placer_cls, oracle_lib):
    """Posting lists with non-contiguous branch sets; read lengths 10..3000
    (several 64-character tiles, lists longer than one wave step)."""
    tree = synth.make_tree(300, seed=2)
    db = synth.make_db(tree.num_nodes, kmer_size=8, seed=3, p_present=0.8, scattered=True)
    rng = np.random.default_rng(9)
    reads = ["".join(rng.choice(list("ACGT"), size=int(n)))
             for n in rng.integers(8, 3000, size=600)]
    data, offs = synth.pack_reads(reads)
    _compare(placer_cls, oracle_lib, db, data, offs)


def test_keep_parameters(placer_cls, oracle_lib, small_case):
    _, db = small_case
    rng = np.random.default_rng(5)
    data, offs = synth.pack_reads(mixed_reads(rng, 500, db.kmer_size))
    for keep_at_most, keep_factor in [(1, 0.01), (3, 0.5), (7, 0.0), (20, 0.001), (64, 0.0)]:
        _compare(placer_cls, oracle_lib, db, data, offs, keep_at_most=keep_at_most,
                 keep_factor=keep_factor)


def test_underflowing_scores_zero_lwr(placer_cls, oracle_lib):
    """Very long reads: 10^score underflows double -> score_sum == 0 -> every LWR is 0
    and nothing is filtered (place.cpp:243-251).  The 40000-base read has more k-mers than
    a 16-bit count holds: place() must switch to the wide kernel by itself."""
    tree = synth.make_tree(40, seed=4)
    db = synth.make_db(tree.num_nodes, kmer_size=6, seed=6, p_present=0.3)
    rng = np.random.default_rng(1)
    reads = ["".join(rng.choice(list("ACGT"), size=n)) for n in (4000, 9000, 20000, 40000)]
    data, offs = synth.pack_reads(reads)
    orc = oracle_lib.Oracle.from_synth(db)
    ref = orc.place(data, offs)
    assert (ref[0]["lwr"][-1] == 0).all() and ref[1][-1] == 7
    with placer_cls.from_synth(db) as pl:
        got = pl.place_packed(data, offs)
    assert_rows_match(*got, *ref)


def test_amino_k4(placer_cls, oracle_lib):
    """20-state encoder (epik-aa), incl. ambiguous B/Z/J/X and invalid '*'."""
    tree = synth.make_tree(30, seed=8)
    db = synth.make_db(tree.num_nodes, states="amino", kmer_size=4, seed=12, p_present=0.4,
                       lognormal=(1.5, 1.0))
    rng = np.random.default_rng(2)
    reads = []
    for i in range(800):
        alpha = alphabet.AMINO_STATES if i % 3 else alphabet.AMINO_STATES + "BZJX*"
        reads.append("".join(rng.choice(list(alpha), size=int(rng.integers(4, 400)))))
    data, offs = synth.pack_reads(reads)
    _compare(placer_cls, oracle_lib, db, data, offs)


def test_amino_k6_sparse(placer_cls, oracle_lib):
    """A sparse protein database (64 M codes, 1 % of them with a list: what the filtered layout is
    chosen for), reads of 300 residues with a few ambiguous ones."""
    tree = synth.make_tree(100, seed=18)
    db = synth.make_db(tree.num_nodes, states="amino", kmer_size=6, seed=19, p_present=0.01,
                       lognormal=(2.0, 1.0))
    rng = np.random.default_rng(20)
    reads = []
    for i in range(1500):
        alpha = alphabet.AMINO_STATES if i % 4 else alphabet.AMINO_STATES + "BZX"
        reads.append("".join(rng.choice(list(alpha), size=300)))
    data, offs = synth.pack_reads(reads)
    _compare(placer_cls, oracle_lib, db, data, offs)


def test_placer_place_mirrors_reference_contract(placer_cls, oracle_lib, small_case):
    """Placer.place(): duplicate sequences are placed once and carry all their
    headers (place.cpp:73-81); lengths are joined per branch (place.cpp:110-123)."""
    tree, db = small_case
    records = [("r1", "ACGTACGTAC"), ("r2", "TTTTACGTAA"), ("r3", "ACGTACGTAC"), ("r4", "GG")]
    with placer_cls.from_synth(db, tree) as pl:
        out = pl.place(records)
    assert out.sequence_map["ACGTACGTAC"] == ["r1", "r3"]
    assert len(out.placed_seqs) == 3
    first = out.placed_seqs[0]
    assert first.sequence == "ACGTACGTAC" and first.placements
    for p in first.placements:
        if p.count:
            assert p.distal_length == pytest.approx(tree.branch_length[p.branch_id] / 2)
    assert out.placed_seqs[2].placements == []      # shorter than k


def test_many_ties_overflow_candidate_buffer(placer_cls, oracle_lib):
    """Every posting has the same score and every list covers all branches, so all
    N=301 corrected scores tie: more candidates than the kernel's LDS candidate
    buffer (192) -> the slow repeated-selection path; ties resolve by branch asc."""
    tree = synth.make_tree(151, seed=3)
    n = tree.num_nodes
    k, sigma = 4, 4
    num_keys = sigma ** k
    offsets = (np.arange(num_keys + 1, dtype=np.uint64) * np.uint64(n))
    values = np.zeros(num_keys * n, dtype=synth.PKDB_VALUE)
    values["branch"] = np.tile(np.arange(n, dtype=np.uint32), num_keys)
    values["score"] = np.float32(-1.25)
    db = synth.SynthDB(states="nucl", kmer_size=k, omega=1.5, num_branches=n, offsets=offsets,
                       values=values)
    data, offs = synth.make_reads(300, 40, seed=5)
    for keep_at_most, keep_factor in [(7, 0.01), (64, 0.0)]:
        _compare(placer_cls, oracle_lib, db, data, offs, keep_at_most=keep_at_most,
                 keep_factor=keep_factor)


def test_large_tree_short_reads_take_the_8_bit_counts(placer_cls, oracle_lib, db_layout):
    """N = 4199: 8-bit counts put more waves on a CU than 16-bit ones, and reads of up to 255 k-mers
    fit them -- place() switches by itself (ambiguous reads included: their "seen" flags move to a
    bitmap); a longer read in the batch switches it back."""
    tree = synth.make_tree(2100, seed=31)
    db = synth.make_db(tree.num_nodes, kmer_size=8, seed=32, p_present=0.5, lognormal=(4.0, 1.5))
    rng = np.random.default_rng(33)
    reads = mixed_reads(rng, 1500, db.kmer_size, max_len=240)
    data, offs = synth.pack_reads(reads)
    orc = oracle_lib.Oracle.from_synth(db)
    with placer_cls.from_synth(db) as pl:
        got = pl.place_packed(data, offs)
        narrow = pl.launch_info()
        assert_rows_match(*got, *orc.place(data, offs, num_threads=0))
        data2, offs2 = synth.pack_reads(reads[:200] + ["ACGT" * 100])
        got2 = pl.place_packed(data2, offs2)
        assert_rows_match(*got2, *orc.place(data2, offs2, num_threads=0))
        if not db_layout.startswith("team"):
            assert narrow["lds_bytes_per_block"] < pl.launch_info()["lds_bytes_per_block"] * narrow["waves_per_block"] / pl.launch_info()["waves_per_block"]


@pytest.mark.parametrize("forced", ["2", "0"])
def test_place_widens_forced_counts_instead_of_marking_reads(placer_cls, oracle_lib, small_case, forced, monkeypatch):
    """EPIK_AMD_WIDE_COUNTS forces a count width for experiments; a read with more k-mers than that width holds must
    still be placed by the host entry point (its consumers take n_rows as a row count: the mark of the
    device-pointer entry points, EPIK_AMD_ROWS_COUNTS_TOO_NARROW, must never come out of epik_amd_placer_place)."""
    from epik_amd import capi
    _, db = small_case
    monkeypatch.setenv("EPIK_AMD_WIDE_COUNTS", forced)   # 8-bit counts (255 k-mers) / 16-bit (32767)
    rng = np.random.default_rng(21)
    long_read = "".join(rng.choice(list("ACGT"), size=40_000 if forced == "0" else 700))
    data, offs = synth.pack_reads(["ACGTACGTAC", long_read, "ACGTTGCA" * 4])
    ref = oracle_lib.Oracle.from_synth(db).place(data, offs, num_threads=0)
    with placer_cls.from_synth(db) as pl:
        got = pl.place_packed(data, offs)
        assert int(got[1].max()) <= pl.keep_at_most and capi.ROWS_COUNTS_TOO_NARROW not in got[1]
        # ... while a device-pointer launch with counts the caller chose too narrow says so, per read
    assert_rows_match(*got, *ref)


def test_reads_cut_from_references_land_in_the_same_rows(placer_cls, oracle_lib):
    """A database built like a real one -- the k-mers of a reference all carry lists over the reference's clade
    (synth.make_clade_db) -- and reads cut from the references: k-mer after k-mer adds into the SAME few dozen rows,
    so consecutive chunks of the stream update the same LDS cells back to back (the order of a row's float32 adds is
    what keeps the score bits; make_db spreads a read's lists over the tree and rarely produces that) and a row's
    count reaches the number of k-mers of the read.  Reads of 150 and of 300 letters (8-bit counts hold 255 k-mers:
    the longer reads take the 16-bit kernels), a few with ambiguous letters."""
    db, refs, _ = synth.make_clade_db(999, n_refs=80, ref_length=700, seed=5)
    d1, o1 = synth.make_clade_reads(refs, 1500, 150, seed=6)
    d2, o2 = synth.make_clade_reads(refs, 500, 300, substitutions=0.03, seed=7)
    reads = [bytes(d1[int(o1[i]):int(o1[i + 1])]) for i in range(1500)] + [bytes(d2[int(o2[i]):int(o2[i + 1])]) for i in range(500)]
    for i in range(0, len(reads), 50):  # an ambiguous letter in the middle of every fiftieth read
        r = bytearray(reads[i])
        r[len(r) // 2] = ord("N") if i % 100 else ord("R")
        reads[i] = bytes(r)
    data, offs = synth.pack_reads(reads)
    orc = oracle_lib.Oracle.from_synth(db)
    ref = orc.place(data, offs, num_threads=0)
    with placer_cls.from_synth(db) as pl:
        got = pl.place_packed(data, offs)
    assert_rows_match(*got, *ref)
    # the workload is what it claims: the best row of a typical read was hit by most of its k-mers
    top_counts = ref[2][:1500, 0]
    assert np.median(top_counts) > 100, np.median(top_counts)
