"""K-mer-space shard of the database (BASELINE configs[4]) on one GPU: the
accumulate / finish halves of the kernel through the C ABI.  With one shard they must give
exactly the rows of the one-pass kernel; with several shards (emulated here by several placers
on the one device, partial vectors added in rank order as `place_kmer_sharded` does) the
float32 sums are reordered, so scores are compared to float32 rounding and like-weight-ratios
to the 1e-5 bar."""
import numpy as np
import pytest

from conftest import assert_rows_match, mixed_reads, select_kernel
from epik_amd import alphabet, dist as edist, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["paired", "compact", "team4", "team8", "team2", "team4-sparse", "team4-classic", "team4-smallpool", "paired-fewblocks", "team4-fewblocks"])
def db_layout(request, monkeypatch):
    """The halves of a sharded placement on the one-wavefront kernels, and on the team kernels as front +
    streaming (+ merge) kernels, as team_place_kernel alone (-classic), and mixed (-smallpool)."""
    select_kernel(monkeypatch, request.param)
    return request.param


def _case():
    tree = synth.make_tree(120, seed=21)
    db = synth.make_db(tree.num_nodes, kmer_size=8, seed=22, p_present=0.7)
    rng = np.random.default_rng(23)
    reads = mixed_reads(rng, 1500, db.kmer_size, max_len=260)
    reads += ["ACG", "", "A" * 8]
    return db, synth.pack_reads(reads)


def test_one_shard_is_the_one_pass_kernel(gpu_available):
    assert gpu_available
    import torch
    from epik_amd.placer import Placer
    db, (data, offs) = _case()
    dev = torch.device("cuda", 0)
    slot, per = edist.amb_slots(data, offs, alphabet.char_class_table(db.states), 1)
    assert per > 100, "the case must hold reads with ambiguous characters"
    with Placer.from_synth(db) as pl:
        ref = pl.place_packed(data, offs)
        accumulate, finish = edist.kmer_sharded_gpu_fns(pl, data, offs, dev)
        # without slots the shard scores its ambiguous k-mers itself; with slots they are recorded by
        # accumulate and added by finish, after the exact scores as in the one-pass loop: same bits
        got = edist.place_kmer_sharded(accumulate, finish, len(offs) - 1, None)
        got_slots = edist.place_kmer_sharded(accumulate, finish, len(offs) - 1, None, amb_slot=slot, amb_per_owner=per)
    assert_rows_match(*got, *ref, lwr_tol=0.0)
    assert_rows_match(*got_slots, *ref, lwr_tol=0.0)


def _emulated_shards(db, data, offs, shards):
    """`shards` placers on the one device; their partial vectors added in rank order and their
    ambiguous-key records combined as `place_kmer_sharded` does."""
    import torch
    from epik_amd.placer import Placer
    n = len(offs) - 1
    dev = torch.device("cuda", 0)
    slot, per = edist.amb_slots(data, offs, alphabet.char_class_table(db.states), 1)
    placers = [Placer.from_synth(db, shard_index=g, shard_count=shards) for g in range(shards)]
    try:
        fns = [edist.kmer_sharded_gpu_fns(p, data, offs, dev) for p in placers]
        parts = [accumulate(n, slot, per) for accumulate, _ in fns]
        total_s, total_c = parts[0][0].clone(), parts[0][1].clone()
        for s, c, _, _ in parts[1:]:  # rank order, like place_kmer_sharded
            total_s += s
            total_c += c
        avg = edist.combine_amb(torch.stack([p[2] for p in parts]), torch.stack([p[3] for p in parts])) if per else None
        return fns[0][1](0, n, total_s, total_c, slot if per else None, avg)
    finally:
        for p in placers:
            p.close()


def _assert_close_to_oracle(got, ref):
    rows, n_rows, counts = got
    ref_rows, ref_n, ref_counts = ref
    assert np.array_equal(n_rows, ref_n)
    valid = np.arange(rows.shape[1])[None, :] < ref_n[:, None]
    same = rows["branch"][valid] == ref_rows["branch"][valid]
    # a reordered float32 sum may swap two rows whose scores agree to rounding; nothing else may differ
    assert same.mean() > 0.999
    np.testing.assert_allclose(rows["score"][valid], ref_rows["score"][valid], rtol=2e-6, atol=0)
    assert np.abs(rows["lwr"][valid] - ref_rows["lwr"][valid]).max() <= 1e-5
    assert np.array_equal(counts[valid][same], ref_counts[valid][same])


@pytest.mark.parametrize("shards", [2, 3])
def test_emulated_shards_match_the_oracle(gpu_available, oracle_lib, shards):
    """Ambiguous reads included: the first-ambiguous-key rule (place.cpp:385-388) holds over all shards."""
    assert gpu_available
    db, (data, offs) = _case()
    got = _emulated_shards(db, data, offs, shards)
    _assert_close_to_oracle(got, oracle_lib.Oracle.from_synth(db).place(data, offs, num_threads=0))


@pytest.mark.parametrize("shards", [2, 3])
def test_n9999_emulated_shards(gpu_available, oracle_lib, shards, db_layout):
    """BASELINE configs[4]: the ~10k-branch tree, k-mer-space sharded."""
    assert gpu_available
    if db_layout == "compact":
        pytest.skip("one layout of the one-wavefront kernel is enough at this size")
    tree = synth.make_tree(5000, seed=42)
    db = synth.make_db(tree.num_nodes, kmer_size=8, seed=47, p_present=0.6, lognormal=(3.5, 1.7))
    rng = np.random.default_rng(50)
    reads = mixed_reads(rng, 300, db.kmer_size, max_len=151)
    reads += ["".join(rng.choice(list("ACGT"), size=150)) for _ in range(500)] + ["ACGTN" * 70]
    data, offs = synth.pack_reads(reads)
    got = _emulated_shards(db, data, offs, shards)
    _assert_close_to_oracle(got, oracle_lib.Oracle.from_synth(db).place(data, offs, num_threads=0))


@pytest.mark.parametrize("worker", ["dist_worker.py", "dist_worker_kmer.py"])
def test_two_ranks_on_the_one_gpu(gpu_available, oracle_lib, worker, db_layout):
    """Both multi-GPU modes (reads sharded / k-mer space sharded) as two real processes sharing
    device 0, gloo for the rendezvous and the exchange (RCCL needs one device per rank)."""
    assert gpu_available
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "tests", worker)]
    env = dict(os.environ, OMP_NUM_THREADS="1", EPIK_AMD_DIST_GPU="1")
    if db_layout == "team8":
        pytest.skip("team4 covers the team kernel here")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "ok: world=2" in out.stdout


# ---- partial LISTS (include/epik_amd.h): the same halves, only the touched rows per (read, slice) ----------
def _lists_or_skip(pl, db_layout):
    info = pl.partial_info()
    if not info["lists"]:
        assert not db_layout.startswith("team") or "classic" in db_layout, db_layout
        pytest.skip("dense partial vectors only on this kernel (small tree / one-kernel placement)")
    return info


def _accumulate_lists(pl, d_seqs, d_offs, n, n_parts, slot, per, cap=None, longest=0):
    """One accumulate_lists call into fresh torch buffers; returns (entries, index, part_entries, order, avg)."""
    import torch
    dev = d_seqs.device
    pl.choose_counts(longest)
    info = pl.partial_info()
    S, eb, N = info["slices"], info["entry_bytes"], info["num_branches"]
    per_part = -(-n // n_parts)
    cap = int(cap if cap is not None else n * 4 * N // 3 + 1024)
    entries = torch.full((max(cap, 1) * eb,), 0xAB, dtype=torch.uint8, device=dev)
    index = torch.zeros((per_part * n_parts, S, 2), dtype=torch.int32, device=dev)
    part_entries = torch.zeros(n_parts, dtype=torch.int64, device=dev)
    d_slot = order = avg = None
    if per:
        d_slot = torch.from_numpy(np.ascontiguousarray(slot, dtype=np.int32)).to(dev)
        order = torch.full((per, N), -1, dtype=torch.int32, device=dev)
        avg = torch.zeros((per, N), dtype=torch.float32, device=dev)
    pl.accumulate_lists_device(d_seqs.data_ptr(), d_offs.data_ptr(), n, n_parts, entries.data_ptr(), cap, index.data_ptr(),
                               part_entries.data_ptr(), 0, d_amb_slot=d_slot.data_ptr() if per else 0,
                               d_amb_order=order.data_ptr() if per else 0, d_amb_avg=avg.data_ptr() if per else 0)
    torch.cuda.synchronize()
    return entries, index, part_entries.cpu().numpy(), order, avg


def _finish_lists(pl, d_offs, begin, end, entries, index, slot, avg):
    import torch
    from epik_amd import capi
    m, keep, dev = end - begin, pl.keep_at_most, d_offs.device
    d_rows = torch.zeros(m * keep * 2, dtype=torch.float64, device=dev)
    d_n = torch.zeros(m, dtype=torch.int32, device=dev)
    d_kc = torch.zeros(m * keep, dtype=torch.int32, device=dev)
    d_slot = torch.from_numpy(np.ascontiguousarray(slot, dtype=np.int32)).to(dev) if slot is not None else None
    pl.finish_lists_device(d_offs.data_ptr() + 8 * begin, m, [e.data_ptr() for e in entries], [x.data_ptr() for x in index],
                           d_rows.data_ptr(), d_n.data_ptr(), d_kc.data_ptr(), 0,
                           d_amb_slot=d_slot.data_ptr() if d_slot is not None else 0,
                           d_amb_avg=avg.data_ptr() if avg is not None else 0)
    torch.cuda.synchronize()
    return (d_rows.cpu().numpy().view(capi.PLACEMENT).reshape(m, keep), d_n.cpu().numpy().view(np.uint32),
            d_kc.cpu().numpy().view(np.uint32).reshape(m, keep))


def _check_lists(entries, index, part_entries, info, n, n_parts):
    """What the header promises about a shard's lists: every list inside its part, rows below slice_rows and at
    most once per list, no list marked as overflowed."""
    eb, S = info["entry_bytes"], info["slices"]
    idx = index.cpu().numpy().view(np.uint32).reshape(-1, S, 2)
    raw = entries.cpu().numpy()
    per = -(-n // n_parts)
    part_first = np.concatenate([[0], np.cumsum(part_entries)])
    assert (idx[:n, :, 1] != 0xFFFFFFFF).all()
    for r in range(0, n, max(1, n // 97)):
        part = r // per
        for s in range(S):
            first, count = int(idx[r, s, 0]), int(idx[r, s, 1])
            assert first + count <= int(part_entries[part])
            lo = (int(part_first[part]) + first) * eb
            words = raw[lo:lo + count * eb].view(np.uint32).reshape(count, eb // 4)
            rows = words[:, 1] & 0xFFFF if eb == 8 else words[:, 1]
            cnts = words[:, 1] >> 16 if eb == 8 else words[:, 2]
            assert (rows < info["slice_rows"]).all() and len(np.unique(rows)) == count and (cnts > 0).all()


def test_lists_one_shard_is_the_one_pass_kernel(gpu_available, db_layout):
    assert gpu_available
    import torch
    from epik_amd.placer import Placer
    db, (data, offs) = _case()
    dev = torch.device("cuda", 0)
    n = len(offs) - 1
    slot, per = edist.amb_slots(data, offs, alphabet.char_class_table(db.states), 1)
    with Placer.from_synth(db) as pl:
        info = _lists_or_skip(pl, db_layout)
        ref = pl.place_packed(data, offs)
        d_seqs = torch.from_numpy(data).to(dev)
        d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
        longest = int(np.diff(offs.astype(np.int64)).max())
        # without slots (the shard adds its ambiguous k-mers itself) and with them; one part and three
        for use_slots, n_parts in ((False, 1), (True, 1), (True, 3)):
            e, ix, pe, order, avg = _accumulate_lists(pl, d_seqs, d_offs, n, n_parts, slot, per if use_slots else 0, longest=longest)
            _check_lists(e, ix, pe, info, n, n_parts)
            part_first = np.concatenate([[0], np.cumsum(pe)])
            per_part = -(-n // n_parts)
            got = []
            for r in range(n_parts):
                b, en = min(n, r * per_part), min(n, (r + 1) * per_part)
                if en == b:
                    continue
                got.append(_finish_lists(pl, d_offs, b, en, [e[int(part_first[r]) * info["entry_bytes"]:]],
                                         [ix[r * per_part:]], slot[b:en] if use_slots else None, avg))
            got = tuple(np.concatenate([g[i] for g in got]) for i in range(3))
            assert_rows_match(*got, *ref, lwr_tol=0.0)


def test_lists_overflow_is_reported_not_written(gpu_available, db_layout):
    """A buffer too small for the parts: the lists that find no room are marked, nothing is written past
    the buffer, and d_part_entries still says what the parts need."""
    assert gpu_available
    import torch
    from epik_amd.placer import Placer
    db, (data, offs) = _case()
    dev = torch.device("cuda", 0)
    n = len(offs) - 1
    with Placer.from_synth(db) as pl:
        info = _lists_or_skip(pl, db_layout)
        d_seqs = torch.from_numpy(data).to(dev)
        d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
        longest = int(np.diff(offs.astype(np.int64)).max())
        _, _, need, _, _ = _accumulate_lists(pl, d_seqs, d_offs, n, 2, None, 0, longest=longest)
        cap = int(need.sum()) // 3
        e, ix, pe, _, _ = _accumulate_lists(pl, d_seqs, d_offs, n, 2, None, 0, cap=cap, longest=longest)
        assert np.array_equal(pe, need)
        counts = ix.cpu().numpy().view(np.uint32)[:n, :, 1]
        assert (counts == 0xFFFFFFFF).any() and (counts != 0xFFFFFFFF).any()
        assert (e.cpu().numpy()[cap * info["entry_bytes"]:] == 0xAB).all()


def _emulated_shards_lists(db, data, offs, shards, n_parts):
    """`shards` placers on the one device leave partial lists for `n_parts` finishers; every part is finished
    from the shards' lists in shard order, the ambiguous records combined as place_kmer_sharded_lists does."""
    import torch
    from epik_amd.placer import Placer
    n = len(offs) - 1
    dev = torch.device("cuda", 0)
    slot, per = edist.amb_slots(data, offs, alphabet.char_class_table(db.states), 1)
    placers = [Placer.from_synth(db, shard_index=g, shard_count=shards) for g in range(shards)]
    try:
        info = placers[0].partial_info()
        if not info["lists"]:
            pytest.skip("dense partial vectors only on this kernel")
        # (the geometry of the lists depends on the tree alone: the same on every shard)
        assert all((p.partial_info()["slices"], p.partial_info()["slice_rows"]) == (info["slices"], info["slice_rows"]) for p in placers)
        d_seqs = torch.from_numpy(data).to(dev)
        d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
        longest = int(np.diff(offs.astype(np.int64)).max())
        parts = [_accumulate_lists(p, d_seqs, d_offs, n, n_parts, slot, per, longest=longest) for p in placers]
        info = placers[0].partial_info()  # (the entry format follows the count width chosen for this batch)
        for e, ix, pe, _, _ in parts:
            _check_lists(e, ix, pe, info, n, n_parts)
        avg = edist.combine_amb(torch.stack([p[3] for p in parts]), torch.stack([p[4] for p in parts])) if per else None
        per_part, eb = -(-n // n_parts), info["entry_bytes"]
        got = []
        for r in range(n_parts):
            b, en = min(n, r * per_part), min(n, (r + 1) * per_part)
            if en == b:
                continue
            entries = [e[int(np.concatenate([[0], np.cumsum(pe)])[r]) * eb:] for e, _, pe, _, _ in parts]
            index = [ix[r * per_part:] for _, ix, _, _, _ in parts]
            got.append(_finish_lists(placers[r % shards], d_offs, b, en, entries, index, slot[b:en] if per else None, avg))
        return tuple(np.concatenate([g[i] for g in got]) for i in range(3))
    finally:
        for p in placers:
            p.close()


@pytest.mark.parametrize("shards,n_parts", [(2, 2), (3, 1), (3, 3)])
def test_lists_emulated_shards_match_the_oracle_and_the_dense_exchange(gpu_available, oracle_lib, shards, n_parts, db_layout):
    """The lists are the dense vectors without their zeros: finish adds them in shard order, so the rows are
    bit for bit those of the dense exchange's rank-order sum."""
    assert gpu_available
    if not db_layout.startswith("team") or "classic" in db_layout:
        pytest.skip("dense partial vectors only on this kernel")
    db, (data, offs) = _case()
    got = _emulated_shards_lists(db, data, offs, shards, n_parts)
    _assert_close_to_oracle(got, oracle_lib.Oracle.from_synth(db).place(data, offs, num_threads=0))
    assert_rows_match(*got, *_emulated_shards(db, data, offs, shards), lwr_tol=0.0)


@pytest.mark.parametrize("shards", [2, 3])
def test_n9999_lists_emulated_shards(gpu_available, oracle_lib, shards, db_layout):
    """BASELINE configs[4] with partial lists, a read of more than 32767 k-mers among them (32-bit counts,
    16-byte entries: the dense vectors' uint16 counts could not hold it)."""
    assert gpu_available
    if not db_layout.startswith("team") or "classic" in db_layout:
        pytest.skip("dense partial vectors only on this kernel")
    tree = synth.make_tree(5000, seed=42)
    db = synth.make_db(tree.num_nodes, kmer_size=8, seed=47, p_present=0.6, lognormal=(3.5, 1.7))
    rng = np.random.default_rng(50)
    reads = mixed_reads(rng, 300, db.kmer_size, max_len=151)
    reads += ["".join(rng.choice(list("ACGT"), size=150)) for _ in range(500)] + ["ACGTN" * 70]
    data, offs = synth.pack_reads(reads)
    got = _emulated_shards_lists(db, data, offs, shards, shards)
    _assert_close_to_oracle(got, oracle_lib.Oracle.from_synth(db).place(data, offs, num_threads=0))
    if shards == 2 and db_layout == "team4":
        reads = reads[:40] + ["".join(rng.choice(list("ACGT"), size=40_000))]
        data, offs = synth.pack_reads(reads)
        got = _emulated_shards_lists(db, data, offs, shards, shards)
        _assert_close_to_oracle(got, oracle_lib.Oracle.from_synth(db).place(data, offs, num_threads=0))


@pytest.mark.parametrize("shards,chunk,margin", [(1, "0", ""), (1, "halves", ""), (2, "300", ""), (3, "97", "0.2")])
def test_place_sharded_native(gpu_available, oracle_lib, shards, chunk, margin, db_layout, monkeypatch):
    """epik_amd_placer_place_sharded: the whole k-mer-space-sharded placement inside the library (what
    epik-dna --db-shard calls) -- chunks of the batch, peer copies of the parts, the exchange of a chunk under the
    accumulate of the next; here all handles on the one device, small chunks, and (margin 0.2) the first
    capacity too small: the overflow round."""
    assert gpu_available
    if not db_layout.startswith("team") or "classic" in db_layout:
        pytest.skip("dense partial vectors only on this kernel")
    from epik_amd.placer import Placer
    monkeypatch.delenv("EPIK_AMD_SHARD_HALVES", raising=False)
    if chunk == "halves":   # one handle takes the one-pass placement by itself: here through accumulate + finish all the same
        monkeypatch.setenv("EPIK_AMD_SHARD_HALVES", "1")
    elif chunk != "0":
        monkeypatch.setenv("EPIK_AMD_SHARD_CHUNK", chunk)
    if margin:
        monkeypatch.setenv("EPIK_AMD_SHARD_MARGIN", margin)
    db, (data, offs) = _case()
    placers = [Placer.from_synth(db, shard_index=g, shard_count=shards) for g in range(shards)]
    try:
        got = Placer.place_sharded(placers, data, offs)
        ref = oracle_lib.Oracle.from_synth(db).place(data, offs, num_threads=0)
        if shards == 1:
            assert_rows_match(*got, *ref)
        else:
            _assert_close_to_oracle(got, ref)
            # bit for bit what the lists give when the test drives the halves itself
            assert_rows_match(*got, *_emulated_shards_lists(db, data, offs, shards, shards), lwr_tol=0.0)
    finally:
        for p in placers:
            p.close()


def test_place_sharded_native_n9999(gpu_available, oracle_lib, db_layout, monkeypatch):
    assert gpu_available
    if db_layout != "team4":
        pytest.skip("one team geometry is enough at this size")
    from epik_amd.placer import Placer
    monkeypatch.setenv("EPIK_AMD_SHARD_CHUNK", "500")
    tree = synth.make_tree(5000, seed=42)
    db = synth.make_db(tree.num_nodes, kmer_size=8, seed=47, p_present=0.6, lognormal=(3.5, 1.7))
    rng = np.random.default_rng(51)
    reads = mixed_reads(rng, 600, db.kmer_size, max_len=151)
    reads += ["".join(rng.choice(list("ACGT"), size=150)) for _ in range(1200)] + ["ACGTN" * 70, "AC", ""]
    data, offs = synth.pack_reads(reads)
    placers = [Placer.from_synth(db, shard_index=g, shard_count=3) for g in range(3)]
    try:
        got = Placer.place_sharded(placers, data, offs)
    finally:
        for p in placers:
            p.close()
    _assert_close_to_oracle(got, oracle_lib.Oracle.from_synth(db).place(data, offs, num_threads=0))


def test_pipelined_batches_of_different_count_widths(gpu_available, oracle_lib, db_layout):
    """place_kmer_sharded_lists starts batch b + 1 before batch b has crossed and finished, and the count width --
    with it the entry format -- is chosen per batch: a batch of short reads (8-bit counts where that kernel exists),
    one with a read of more than 255 k-mers (16-bit) and one with a read of more than 32 767 (32-bit counts,
    16-byte entries) must each be exchanged and finished with the width they were accumulated with."""
    assert gpu_available
    if not db_layout.startswith("team") or "classic" in db_layout:
        pytest.skip("dense partial vectors only on this kernel")
    import torch
    from epik_amd.placer import Placer
    db, _ = _case()
    rng = np.random.default_rng(77)
    short = ["".join(rng.choice(list("ACGT"), size=int(n))) for n in rng.integers(db.kmer_size, 120, size=150)]
    mid = short[:40] + ["".join(rng.choice(list("ACGT"), size=700))]
    long_ = short[40:70] + ["".join(rng.choice(list("ACGT"), size=40_000))]
    batches = [synth.pack_reads(b) for b in (short, mid, long_, short)]
    with Placer.from_synth(db) as pl:
        engine = edist.ListsGpuEngine(pl, torch.device("cuda", 0))
        widths = []
        accumulate = engine.accumulate
        engine.accumulate = lambda *a, **kw: (lambda parts: (widths.append(parts.entry_bytes), parts)[1])(accumulate(*a, **kw))
        results = list(edist.place_kmer_sharded_lists(engine, batches, None))
    assert widths == [8, 8, 16, 8], widths
    orc = oracle_lib.Oracle.from_synth(db)
    for (data, offs), got in zip(batches, results):
        assert_rows_match(*got, *orc.place(data, offs, num_threads=0))


def test_dense_partials_mark_reads_of_more_than_65535_kmers(gpu_available, db_layout):
    """The dense partial vectors count in uint16: a read of more k-mers must come back marked
    (EPIK_AMD_ROWS_COUNTS_TOO_NARROW), never with wrapped counts -- whatever the LDS counts of the launch hold
    (32-bit for such a read).  The partial lists have no such limit (test_n9999_lists_emulated_shards)."""
    assert gpu_available
    import torch
    from epik_amd import capi
    from epik_amd.placer import Placer
    db, _ = _case()
    rng = np.random.default_rng(31)
    data, offs = synth.pack_reads(["".join(rng.choice(list("ACGT"), size=70_000)), "ACGTACGTACGTAAC"])
    dev = torch.device("cuda", 0)
    with Placer.from_synth(db) as pl:
        accumulate, finish = edist.kmer_sharded_gpu_fns(pl, data, offs, dev)
        rows, n_rows, _ = edist.place_kmer_sharded(accumulate, finish, 2, None)
    assert int(n_rows[0]) == capi.ROWS_COUNTS_TOO_NARROW and 1 <= int(n_rows[1]) <= 7
