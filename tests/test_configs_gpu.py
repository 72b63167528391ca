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
from epik_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amino_k7(gpu_available):
    assert gpu_available, "pytest -m gpu needs a HIP device (no CPU fallback exists)"
    tree = synth.make_tree(500, seed=42)                       # N = 999
    db = synth.make_sparse_db(tree.num_nodes, states="amino", kmer_size=7, p_present=0.0026, seed=43)
    data, offs = synth.reads_hitting(db, 3000, 300, hit_rate=0.25, seed=46, dirty="BZXJ*")
    return db, data, offs


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


@pytest.fixture(scope="module")
def large_tree(gpu_available):
    assert gpu_available
    tree = synth.make_tree(5000, seed=42)                      # N = 9 999
    db = synth.make_db(tree.num_nodes, kmer_size=8, seed=47, p_present=0.6, lognormal=(3.5, 1.7))
    return tree, db


@pytest.mark.parametrize("counts", ["auto", "0", "1", "2"])
@pytest.mark.parametrize("layout", ["default", "paired", "compact", "team4", "team8", "team4-classic", "team4-smallpool"])
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


@pytest.mark.parametrize("kernel", ["wave", "team4", "team8", "team4-classic", "team4-smallpool"])
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
