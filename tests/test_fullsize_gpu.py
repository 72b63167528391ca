"""BASELINE configs[1] at its full size (N=999, k=10, 1 M x 150 bp reads, the bench.py workload), a mid-size
tree (N=2 999) and the configs[4] tree (N=9 999, the team kernels) on 200-400 k reads: too many reads for the oracle, so the
whole batch goes through size-independent properties and a random sample of it through the oracle,
bit for bit."""
import numpy as np
import pytest

from conftest import assert_rows_match
from epik_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[(500, 1_000_000, False), (1500, 400_000, False), (5000, 300_000, False), (5000, 200_000, True)],
                ids=["n999", "n2999", "n9999", "n9999-clades"])
def full_case(gpu_available, request):
    """n999: one wavefront per read; n2999: two slices per pass, the lean streaming kernel; n9999: four slices, the
    wide one -- dense and touched-quad epilogues side by side on SURVEY 8d's random lists, touched-quad and empty
    slices only on lists over the clades of references (synth.make_clade_db, reads cut from the references)."""
    assert gpu_available, "pytest -m gpu needs a HIP device (no CPU fallback exists)"
    from epik_amd.placer import Placer
    leaves, N_READS, clades = request.param
    tree = synth.make_tree(leaves, seed=42)
    if clades:
        db, refs, _ = synth.make_clade_db(tree.num_nodes, kmer_size=10, seed=47)
        data, offs = synth.make_clade_reads(refs, N_READS, 150, seed=48)
    else:
        db = synth.make_db(tree.num_nodes, kmer_size=10, seed=43)
        data, offs = synth.make_reads(N_READS, 150, seed=44)
    with Placer.from_synth(db) as pl:
        first = pl.place_packed(data, offs)
        again = pl.place_packed(data, offs)
        # the same reads in another order: every read is placed on its own, nothing of a wave's
        # previous read may leak into the next one (the LDS vectors are reset per read)
        perm = np.random.default_rng(5).permutation(N_READS)
        shuffled = pl.place_packed(data.reshape(N_READS, 150)[perm].reshape(-1), offs)
    return db, data, offs, first, again, perm, shuffled


def test_rows_are_well_formed(full_case):
    db, _, _, (rows, n_rows, counts), *_ = full_case
    keep = rows.shape[1]
    assert n_rows.min() >= 1 and n_rows.max() <= keep
    valid = np.arange(keep)[None, :] < n_rows[:, None]
    assert rows["branch"][valid].max() < db.num_branches
    score = np.where(valid, rows["score"], -np.inf)
    with np.errstate(invalid="ignore"):  # (-inf) - (-inf) behind a read's last row
        steps = np.diff(score, axis=1)
    assert (steps[valid[:, 1:]] <= 0).all(), "scores must be sorted in descending order"
    lwr = np.where(valid, rows["lwr"], 0.0)
    # (sum_scores is accumulated relative to its largest term in float32, ~1e-7 relative: a read whose reported rows
    # carry all but 1e-4 of the mass -- reads cut from a reference -- may add up to 1 + 1e-7; the bar on a ratio is 1e-5)
    assert (lwr >= 0).all() and (lwr.sum(axis=1) <= 1.0 + 1e-6).all()
    assert (lwr[:, :1] >= lwr).all(), "the best row carries the largest like_weight_ratio"
    assert (lwr[valid] >= 0.01 * np.repeat(lwr[:, 0], n_rows) - 1e-15).all(), "filter_by_ratio (place.cpp:188-199)"
    assert (counts[valid] <= 141).all() and (counts[valid] >= 0).all()
    # equal like_weight_ratio <=> equal score inside a read (both are monotone in the score)
    same_score = (steps == 0) & valid[:, 1:]
    assert (np.diff(lwr, axis=1)[same_score] == 0).all()


def test_idempotent_and_order_independent(full_case):
    _, _, _, first, again, perm, shuffled = full_case
    for a, b in zip(first, again):
        assert a.tobytes() == b.tobytes(), "two runs over the same batch differ"
    for a, b in zip(first, shuffled):
        assert a[perm].tobytes() == b.tobytes(), "a read's rows depend on its position in the batch"


def test_random_sample_matches_the_oracle(full_case, oracle_lib):
    db, data, offs, (rows, n_rows, counts), *_ = full_case
    pick = np.sort(np.random.default_rng(9).choice(len(n_rows), size=4000, replace=False))
    sample, sample_offs = synth.pack_reads([bytes(data[int(offs[i]):int(offs[i + 1])]) for i in pick])
    ref = oracle_lib.Oracle.from_synth(db).place(sample, sample_offs, num_threads=0)
    assert_rows_match(rows[pick], n_rows[pick], counts[pick], *ref)


# ---- BASELINE configs[2]: 100 M x 150 bp reads over 8 GPUs, database replicated: ONE GPU's share -------------------
SHARE_READS = 12_500_000


def _check_rows_in_pieces(rows, n_rows, counts, num_branches, piece=1_000_000):
    """test_rows_are_well_formed's properties, a million reads at a time (12.5 M reads x 7 rows x 16 bytes are 1.4 GB:
    no full-size temporaries)."""
    keep = rows.shape[1]
    for at in range(0, len(n_rows), piece):
        r, n, c = rows[at:at + piece], n_rows[at:at + piece], counts[at:at + piece]
        assert n.min() >= 1 and n.max() <= keep
        valid = np.arange(keep)[None, :] < n[:, None]
        assert r["branch"][valid].max() < num_branches
        score = np.where(valid, r["score"], -np.inf)
        with np.errstate(invalid="ignore"):
            steps = np.diff(score, axis=1)
        assert (steps[valid[:, 1:]] <= 0).all(), "scores must be sorted in descending order"
        lwr = np.where(valid, r["lwr"], 0.0)
        assert (lwr >= 0).all() and (lwr.sum(axis=1) <= 1.0 + 1e-6).all()
        assert (lwr[:, :1] >= lwr).all()
        assert (lwr[valid] >= 0.01 * np.repeat(lwr[:, 0], n) - 1e-15).all(), "filter_by_ratio (place.cpp:188-199)"
        assert (c[valid] <= 141).all()


def test_one_gpus_share_of_configs2(gpu_available, oracle_lib):
    """12.5 M x 150 bp reads (configs[2] / 8: the OpenMP loop of place.cpp:218-230 over one GPU's share of the
    hundred million) on the N = 999 database: once in ONE `epik_amd_placer_place_device` call on device-resident reads
    (what bench.py --gpus 8 does per rank and step at its largest), once through `epik_amd_placer_place` (host
    buffers in chunks).  Both must give the same bytes; the whole batch goes through the property checks, 4 000
    reads of it -- drawn over the whole range, the last read included -- through the oracle."""
    assert gpu_available
    import torch
    from epik_amd import capi
    from epik_amd.placer import Placer
    tree = synth.make_tree(500, seed=42)
    db = synth.make_db(tree.num_nodes, kmer_size=10, seed=43)
    data, offs = synth.make_reads(SHARE_READS, 150, seed=44 + 3)   # (rank 3's seed in bench.py)
    n = SHARE_READS
    with Placer.from_synth(db) as pl:
        keep = pl.keep_at_most
        dev = torch.device("cuda", 0)
        d_seqs = torch.from_numpy(data).to(dev)
        d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
        d_rows = torch.zeros(n * keep * 2, dtype=torch.float64, device=dev)
        d_nrows = torch.zeros(n, dtype=torch.int32, device=dev)
        d_counts = torch.zeros(n * keep, dtype=torch.int32, device=dev)
        pl.choose_counts(150)
        stream = torch.cuda.current_stream()
        pl.place_device(d_seqs.data_ptr(), d_offs.data_ptr(), n, d_rows.data_ptr(), d_nrows.data_ptr(), d_counts.data_ptr(),
                        stream.cuda_stream)
        torch.cuda.synchronize()
        assert pl.last_path() == capi.PATH_WAVE
        rows = d_rows.cpu().numpy().view(capi.PLACEMENT).reshape(n, keep)
        n_rows = d_nrows.cpu().numpy().view(np.uint32)
        counts = d_counts.cpu().numpy().view(np.uint32).reshape(n, keep)
        del d_rows, d_nrows, d_counts, d_seqs, d_offs
        torch.cuda.empty_cache()
        _check_rows_in_pieces(rows, n_rows, counts, db.num_branches)
        # the same reads through the host entry point: pageable buffers in, rows out, chunked by the library
        h_rows, h_n, h_counts = pl.place_packed(data, offs)
    assert h_n.tobytes() == n_rows.tobytes()
    valid = np.arange(keep)[None, :] < n_rows[:, None]
    for field in ("branch", "score", "lwr"):
        assert np.array_equal(h_rows[field][valid].view(np.uint32 if field != "lwr" else np.uint64),
                              rows[field][valid].view(np.uint32 if field != "lwr" else np.uint64)), field
    assert np.array_equal(h_counts[valid], counts[valid])
    del h_rows, h_counts, valid
    pick = np.unique(np.concatenate([np.random.default_rng(9).choice(n, size=3998, replace=False), [0, n - 1]]))
    sample, sample_offs = synth.pack_reads([bytes(data[int(offs[i]):int(offs[i + 1])]) for i in pick])
    ref = oracle_lib.Oracle.from_synth(db).place(sample, sample_offs, num_threads=0)
    assert_rows_match(rows[pick], n_rows[pick], counts[pick], *ref)
