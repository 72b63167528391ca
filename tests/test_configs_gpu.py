"""The BASELINE.json configurations whose shapes select code no other test reaches, each against the
CPU oracle bit for bit through the C ABI:

* configs[3] -- amino k = 7: 20^7 = 1.28 G codes (one bit short of 2^32), a 10 GB lookup table, the
  512 MB presence filter, 300-residue reads with a few B / Z / X;
* configs[4] -- N = 9 999 branches: the per-branch vectors of one read fill a third of a CU's LDS
  (one-pass here; the k-mer-space shard of the same tree is in test_kmer_shard_gpu.py).
"""
import numpy as np
import pytest

from conftest import assert_rows_match, mixed_reads, select_kernel
from epik_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amino_k7(gpu_available):
    assert gpu_available, "pytest -m gpu needs a HIP device (no CPU fallback exists)"
    tree = synth.make_tree(500, seed=42)                       # N = 999
    db = synth.make_sparse_db(tree.num_nodes, states="amino", kmer_size=7, p_present=0.0026, seed=43)
    data, offs = synth.reads_hitting(db, 3000, 300, hit_rate=0.25, seed=46, dirty="BZXJ*")
    return db, data, offs


@pytest.fixture(scope="module")
def amino_k7_placer(amino_k7):
    """One placer on the protein database for the tests that take what create() chooses (the filtered layout): building
    the 10 GB table again for each of them is most of their time."""
    import os
    from epik_amd.placer import Placer
    saved = {v: os.environ.pop(v, None) for v in ("EPIK_AMD_KERNEL", "EPIK_AMD_LAYOUT", "EPIK_AMD_RUNS")}
    try:
        pl = Placer.from_synth(amino_k7[0])
    finally:
        os.environ.update({k: v for k, v in saved.items() if v is not None})
    yield pl
    pl.close()


@pytest.mark.parametrize("layout", ["filtered", "packed"])
def test_amino_k7_matches_the_oracle(amino_k7, oracle_lib, layout, monkeypatch):
    from epik_amd.placer import Placer
    db, data, offs = amino_k7
    monkeypatch.setenv("EPIK_AMD_KERNEL", "wave")
    monkeypatch.setenv("EPIK_AMD_LAYOUT", layout)
    ref = oracle_lib.Oracle.from_synth(db).place(data, offs, num_threads=0)
    assert (ref[2][:, 0] > 0).mean() > 0.9, "the reads must find their planted k-mers"
    with Placer.from_synth(db) as pl:
        got = pl.place_packed(data, offs)
    assert_rows_match(*got, *ref)


def test_amino_k7_sparse_descriptor_needs_no_array_per_code(amino_k7, oracle_lib, monkeypatch):
    """ABI 3: keys[num_present] + offsets[num_present + 1].  The dense form costs 8 bytes per POSSIBLE k-mer
    before a posting is loaded -- 10 GB for amino k = 7 --, the sparse one memory per present key, like the hash map
    behind phylo_kmer_db::search (place.cpp:300).  create() streams from it: beyond the caller's arrays the process
    grows by its staging buffers only (the dense test_build_streams_without_a_host_copy asserts the same)."""
    import threading
    import time
    import psutil
    from epik_amd.placer import Placer
    db, data, offs = amino_k7
    monkeypatch.setenv("EPIK_AMD_KERNEL", "wave")
    monkeypatch.delenv("EPIK_AMD_LAYOUT", raising=False)
    # the sparse form of the same database (built here from the dense arrays the oracle needs anyway)
    lens = np.diff(db.offsets.view(np.int64))
    present = np.nonzero(lens)[0]
    keys = present.astype(np.uint32)
    offsets = np.concatenate([[0], np.cumsum(lens[present])]).astype(np.uint64)
    del lens, present
    assert keys.nbytes + offsets.nbytes < 64 << 20   # against 10 GB of dense offsets
    proc = psutil.Process()
    peak, stop = [0], threading.Event()

    def sample():
        while not stop.is_set():
            peak[0] = max(peak[0], proc.memory_info().rss)
            time.sleep(0.005)

    before = proc.memory_info().rss
    t = threading.Thread(target=sample)
    t.start()
    try:
        pl = Placer(offsets, db.values, keys=keys, states=db.states, kmer_size=db.kmer_size, num_branches=db.num_branches,
                    threshold=db.threshold, log_threshold=db.log_threshold)
    finally:
        stop.set()
        t.join()
    grown = peak[0] - before
    assert grown < 1 << 30, f"create() from the sparse descriptor grew the process by {grown >> 20} MiB"
    with pl:
        got = pl.place_packed(data, offs)
    assert_rows_match(*got, *oracle_lib.Oracle.from_synth(db).place(data, offs, num_threads=0))


def test_amino_k7_100k_reads(amino_k7, amino_k7_placer, oracle_lib, monkeypatch):
    """configs[3] beyond a handful of reads: 100 000 protein reads of 300 residues (B / Z / X / J / * in a quarter of
    them) through the size-independent properties -- well-formed rows, idempotent, independent of the order of
    the batch -- and a random sample of them through the oracle, bit for bit (as test_fullsize_gpu does for DNA)."""
    from epik_amd.placer import Placer
    db, _, _ = amino_k7
    monkeypatch.delenv("EPIK_AMD_KERNEL", raising=False)
    monkeypatch.delenv("EPIK_AMD_LAYOUT", raising=False)   # what create() chooses: the filtered layout
    n = 100_000
    data, offs = synth.reads_hitting(db, n, 300, hit_rate=0.25, seed=52, dirty="BZXJ*")
    pl = amino_k7_placer
    first = pl.place_packed(data, offs)
    again = pl.place_packed(data, offs)
    perm = np.random.default_rng(6).permutation(n)
    shuffled = pl.place_packed(data.reshape(n, 300)[perm].reshape(-1), offs)
    rows, n_rows, counts = first
    keep = rows.shape[1]
    assert n_rows.min() >= 1 and n_rows.max() <= keep
    valid = np.arange(keep)[None, :] < n_rows[:, None]
    assert rows["branch"][valid].max() < db.num_branches
    assert (np.diff(rows["score"], axis=1)[valid[:, 1:]] <= 0).all(), "scores must be sorted in descending order"
    lwr = np.where(valid, rows["lwr"], 0.0)
    assert (lwr >= 0).all() and (lwr.sum(axis=1) <= 1.0 + 1e-9).all() and (lwr[:, :1] >= lwr).all()
    assert (lwr[valid] >= 0.01 * np.repeat(lwr[:, 0], n_rows) - 1e-15).all()     # filter_by_ratio, place.cpp:188-199
    assert (counts[valid] <= 294).all()
    for a, b in zip(first, again):
        assert a.tobytes() == b.tobytes()
    for a, b in zip(first, shuffled):
        assert a[perm].tobytes() == b.tobytes()
    pick = np.sort(np.random.default_rng(10).choice(n, size=2500, replace=False))
    sample, sample_offs = synth.pack_reads([bytes(data[int(offs[i]):int(offs[i + 1])]) for i in pick])
    ref = oracle_lib.Oracle.from_synth(db).place(sample, sample_offs, num_threads=0)
    assert_rows_match(rows[pick], n_rows[pick], counts[pick], *ref)


def test_amino_k7_at_the_stated_size(gpu_available, oracle_lib, monkeypatch):
    """configs[3] at the size BASELINE.md states: p_present 0.0133 of the 1.28 G codes, about 1 G postings (8 GB of
    values, an 18 GB image), handed to create() in the sparse form of the descriptor -- nothing here holds an array
    per possible code, the oracle neither: it searches through its hash map built from the same keys
    (oracle.Oracle.from_sparse).  50 000 reads through the size-independent properties, a sample of them bit for bit."""
    assert gpu_available
    from epik_amd.placer import Placer
    for var in ("EPIK_AMD_KERNEL", "EPIK_AMD_LAYOUT", "EPIK_AMD_RUNS", "EPIK_AMD_FILTER"):
        monkeypatch.delenv(var, raising=False)
    tree = synth.make_tree(500, seed=42)
    db = synth.make_sparse_db(tree.num_nodes, states="amino", kmer_size=7, p_present=0.0133, seed=43, dense=False)
    assert db.keys is not None and 0.9e9 < db.num_entries < 1.1e9 and db.offsets.shape[0] == db.keys.shape[0] + 1
    n = 50_000
    data, offs = synth.reads_hitting(db, n, 300, hit_rate=0.25, seed=53, dirty="BZXJ*")
    with Placer.from_synth(db) as pl:
        rows, n_rows, counts = pl.place_packed(data, offs)
        again = pl.place_packed(data, offs)
    for a, b in zip((rows, n_rows, counts), again):
        assert a.tobytes() == b.tobytes()
    keep = rows.shape[1]
    assert n_rows.min() >= 1 and n_rows.max() <= keep
    valid = np.arange(keep)[None, :] < n_rows[:, None]
    assert rows["branch"][valid].max() < db.num_branches
    assert (np.diff(rows["score"], axis=1)[valid[:, 1:]] <= 0).all(), "scores must be sorted in descending order"
    lwr = np.where(valid, rows["lwr"], 0.0)
    assert (lwr >= 0).all() and (lwr.sum(axis=1) <= 1.0 + 1e-9).all() and (lwr[:, :1] >= lwr).all()
    assert (lwr[valid] >= 0.01 * np.repeat(lwr[:, 0], n_rows) - 1e-15).all()     # filter_by_ratio, place.cpp:188-199
    assert (counts[valid] <= 294).all() and (counts[:, 0] > 0).mean() > 0.9, "the reads must find their planted k-mers"
    pick = np.sort(np.random.default_rng(11).choice(n, size=1500, replace=False))
    sample, sample_offs = synth.pack_reads([bytes(data[int(offs[i]):int(offs[i + 1])]) for i in pick])
    ref = oracle_lib.Oracle.from_sparse(db).place(sample, sample_offs, num_threads=0)
    assert_rows_match(rows[pick], n_rows[pick], counts[pick], *ref)


def test_protein_reads_across_the_underflow_limits(amino_k7, amino_k7_placer, oracle_lib, monkeypatch):
    """sum_scores has three regimes by the size of its largest term 10^ref: relative to it in float32 (ref > -280),
    term by term in double (down to -325, where a double still holds a denormal), and nothing to add at all below that
    (every term is 0 in double: the epilogue skips the loop).  Protein reads of 170 to 350 residues cross both limits
    -- their threshold score is n_kmers * log10(threshold) / 7, from -209 to -438 -- and every one must come out as the
    oracle's, like-weight ratios of 0 and denormal ratios included."""
    from epik_amd.placer import Placer
    db, _, _ = amino_k7
    monkeypatch.delenv("EPIK_AMD_KERNEL", raising=False)
    monkeypatch.delenv("EPIK_AMD_LAYOUT", raising=False)
    lengths = np.repeat(np.arange(170, 351, 2), 12)
    long_data, _ = synth.reads_hitting(db, len(lengths), 350, hit_rate=0.3, seed=1000, dirty="BX")  # (one draw: finding
    long_data = long_data.reshape(len(lengths), 350)                                               # present codes is slow)
    data, offs = synth.pack_reads([bytes(long_data[i, :n]) for i, n in enumerate(lengths)])
    ref = oracle_lib.Oracle.from_synth(db).place(data, offs, num_threads=0)
    best = ref[0]["score"][:, 0]
    assert (best > -280).sum() > 50 and ((best < -280) & (best > -325)).sum() > 50 and (best < -325).sum() > 50, \
        "the reads must fall on all three sides of the two limits"
    lwr0 = ref[0]["lwr"][:, 0]
    assert (lwr0 == 0).any() and (lwr0 > 0).any()
    got = amino_k7_placer.place_packed(data, offs)
    assert_rows_match(*got, *ref)


@pytest.mark.parametrize("n_branches", [1983, 1984, 1985, 2047, 2048, 2049, 2999, 3499, 3501, 3999, 4095, 4097, 4199, 4200, 4201, 5999, 10001, 14999, 19999, 30001, 50001])
def test_tree_sizes_around_the_kernel_thresholds(gpu_available, oracle_lib, n_branches, monkeypatch):
    """What create() picks by itself on either side of its thresholds -- one wavefront per read below N = 1 984, two
    slices per pass up to 4 200, four beyond (db_image.cpp: make_plan, choose_team) -- and around the powers of two
    where the padded row counts step, sizes on either side of the streaming kernel's choice of workgroup (two waves at
    2 999, 3 999 and 5 999, four at 3 499: db_layout.h: stream_block_waves), and the sizes beyond one pass (two, three and more passes by the rule of
    choose_team): every one against the oracle."""
    assert gpu_available
    from epik_amd.placer import Placer
    for var in ("EPIK_AMD_KERNEL", "EPIK_AMD_LAYOUT", "EPIK_AMD_RUNS", "EPIK_AMD_WIDE_COUNTS", "EPIK_AMD_TEAM_SPARSE"):
        monkeypatch.delenv(var, raising=False)
    db = synth.make_db(n_branches, kmer_size=7, seed=90 + n_branches % 7, p_present=0.6, lognormal=(3.5, 1.7))
    rng = np.random.default_rng(n_branches)
    reads = mixed_reads(rng, 600, db.kmer_size, max_len=151)
    reads += ["".join(rng.choice(list("ACGT"), size=150)) for _ in range(900)]
    reads += ["".join(rng.choice(list("ACGT"), size=400))]     # one read beyond the 8-bit counts
    data, offs = synth.pack_reads(reads)
    ref = oracle_lib.Oracle.from_synth(db).place(data, offs, num_threads=0)
    with Placer.from_synth(db) as pl:
        got = pl.place_packed(data, offs)
        short = pl.place_packed(*synth.pack_reads(reads[:-1]))   # ... and the same batch without it (8-bit counts)
    assert_rows_match(*got, *ref)
    assert_rows_match(*short, *(a[:-1] for a in ref))


@pytest.fixture(scope="module")
def large_tree(gpu_available):
    assert gpu_available
    tree = synth.make_tree(5000, seed=42)                      # N = 9 999
    db = synth.make_db(tree.num_nodes, kmer_size=8, seed=47, p_present=0.6, lognormal=(3.5, 1.7))
    return tree, db


@pytest.mark.parametrize("counts", ["auto", "0", "1", "2"])
@pytest.mark.parametrize("layout", ["default", "paired", "compact", "team4", "team2x2", "team4-sparse", "team4-dense", "team8", "team4-classic", "team4-smallpool"])
def test_n9999_one_pass(large_tree, oracle_lib, counts, layout, monkeypatch):
    """150 bp reads (141 k-mers: the 8-bit counts apply), with ambiguous and invalid characters in a
    third of them; `auto` lets place() choose, 0 / 1 / 2 force 16- / 32- / 8-bit counts."""
    from epik_amd.placer import Placer
    _, db = large_tree
    if layout != "default":  # default: what create() chooses for this tree (the team kernels)
        select_kernel(monkeypatch, layout)
    if counts != "auto":
        monkeypatch.setenv("EPIK_AMD_WIDE_COUNTS", counts)
    rng = np.random.default_rng(48)
    reads = mixed_reads(rng, 1200, db.kmer_size, max_len=151)
    reads += ["".join(rng.choice(list("ACGT"), size=150)) for _ in range(1800)]
    data, offs = synth.pack_reads(reads)
    ref = oracle_lib.Oracle.from_synth(db).place(data, offs, num_threads=0)
    with Placer.from_synth(db) as pl:
        got = pl.place_packed(data, offs)
    assert_rows_match(*got, *ref)


@pytest.mark.parametrize("kernel", ["wave", "team4", "team4-sparse", "team8", "team4-classic", "team4-smallpool"])
def test_n9999_long_reads_leave_the_8_bit_counts(large_tree, oracle_lib, kernel, monkeypatch):
    """One read of more than 255 k-mers in the batch: place() must not pick the 8-bit counts; and
    reads long enough for several passes over the tiles."""
    from epik_amd.placer import Placer
    _, db = large_tree
    if kernel == "wave":
        monkeypatch.setenv("EPIK_AMD_KERNEL", kernel)
    else:
        select_kernel(monkeypatch, kernel)
    rng = np.random.default_rng(49)
    reads = ["".join(rng.choice(list("ACGT"), size=int(n))) for n in rng.integers(8, 150, size=400)]
    reads += ["".join(rng.choice(list("ACGTN"), size=int(n))) for n in (263, 700, 3000, 40000)]
    data, offs = synth.pack_reads(reads)
    ref = oracle_lib.Oracle.from_synth(db).place(data, offs, num_threads=0)
    with Placer.from_synth(db) as pl:
        got = pl.place_packed(data, offs)
    assert_rows_match(*got, *ref)


@pytest.mark.parametrize("kernel", ["wave", "team4"])
def test_placers_of_different_trees_share_a_process(large_tree, small_case, oracle_lib, kernel, monkeypatch):
    """The dynamic-LDS cap of a kernel is per process: creating a placer for a small tree must not
    lower it under the launches of a live placer for a large one (more than 64 KB per workgroup here)."""
    from epik_amd.placer import Placer
    monkeypatch.setenv("EPIK_AMD_KERNEL", kernel)
    monkeypatch.setenv("EPIK_AMD_WIDE_COUNTS", "1")   # 32-bit counts: 82 KB of LDS per workgroup
    monkeypatch.setenv("EPIK_AMD_STREAM_BLOCK", "4")  # (... of four waves: by itself this geometry takes two, 41 KB)
    _, big = large_tree
    _, small = small_case
    rng = np.random.default_rng(51)
    data, offs = synth.pack_reads(mixed_reads(rng, 400, big.kmer_size, max_len=151))
    sdata, soffs = synth.pack_reads(mixed_reads(rng, 400, small.kmer_size, max_len=100))
    with Placer.from_synth(big) as first:
        before = first.place_packed(data, offs)
        assert first.launch_info()["lds_bytes_per_block"] > 64 << 10
        with Placer.from_synth(small) as second:
            assert_rows_match(*second.place_packed(sdata, soffs),
                              *oracle_lib.Oracle.from_synth(small).place(sdata, soffs, num_threads=0))
            after = first.place_packed(data, offs)
    for a, b in zip(before, after):
        assert a.tobytes() == b.tobytes()
    assert_rows_match(*after, *oracle_lib.Oracle.from_synth(big).place(data, offs, num_threads=0))


def test_release_scratch_and_place_again(large_tree, monkeypatch):
    """A large-tree handle keeps the scratch of its launches (descriptor pool, headers, slice results) and the
    staging of the host entry point; epik_amd_placer_release_scratch gives all of it back -- the device's free
    memory returns to what it was after create() -- and the next placement allocates again and gives the same rows."""
    import torch
    from epik_amd.placer import Placer
    _, db = large_tree
    for var in ("EPIK_AMD_KERNEL", "EPIK_AMD_LAYOUT", "EPIK_AMD_TEAM_FRONT", "EPIK_AMD_TEAM_POOL"):
        monkeypatch.delenv(var, raising=False)
    rng = np.random.default_rng(77)
    data, offs = synth.pack_reads(["".join(rng.choice(list("ACGT"), size=150)) for _ in range(20000)])
    with Placer.from_synth(db) as pl:
        torch.cuda.synchronize()
        free_created = torch.cuda.mem_get_info()[0]
        first = pl.place_packed(data, offs)
        free_used = torch.cuda.mem_get_info()[0]
        assert free_used < free_created - (8 << 20), "the placement must have grown scratch on the device"
        pl.release_scratch()
        assert torch.cuda.mem_get_info()[0] >= free_created - (1 << 20)
        again = pl.place_packed(data, offs)
    for a, b in zip(first, again):
        assert a.tobytes() == b.tobytes()


def test_long_reads_on_the_large_tree_across_the_underflow_limits(large_tree, oracle_lib, monkeypatch):
    """The same three regimes of sum_scores (test_protein_reads_across_the_underflow_limits) through the team kernels,
    where every slice hands the merge its share of the sum: nucleotide reads of 300 to 1 300 letters on the N = 9 999
    tree (k = 8: the threshold score is n_kmers * -4.64 / 8, -170 to -750; the best row lies well above it)."""
    from epik_amd.placer import Placer
    _, db = large_tree
    monkeypatch.delenv("EPIK_AMD_KERNEL", raising=False)   # what create() chooses for this tree: the team kernels
    monkeypatch.delenv("EPIK_AMD_LAYOUT", raising=False)
    rng = np.random.default_rng(77)
    reads = ["".join(rng.choice(list("ACGT"), size=length)) for length in range(300, 1301, 8) for _ in range(4)]
    data, offs = synth.pack_reads(reads)
    ref = oracle_lib.Oracle.from_synth(db).place(data, offs, num_threads=0)
    best = ref[0]["score"][:, 0]
    assert (best > -280).sum() > 50 and ((best < -280) & (best > -325)).sum() > 20 and (best < -325).sum() > 50, \
        (int((best > -280).sum()), int(((best < -280) & (best > -325)).sum()), int((best < -325).sum()))
    with Placer.from_synth(db) as pl:
        got = pl.place_packed(data, offs)
        assert pl.last_path() != capi.PATH_WAVE
    assert_rows_match(*got, *ref)
