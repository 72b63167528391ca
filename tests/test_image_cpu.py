"""The host-side image builder (epik_amd/csrc/db_image.cpp) without a device: every layout byte for
byte against a plain numpy / Python restatement of its description in place_kernel.hip /
team_kernel.hip, the plan create() would make, and the memory the streaming build needs."""
import os
import threading
import time

import numpy as np
import psutil
import pytest

from epik_amd import capi, placer as eplacer, synth

LINE = 128


def _lists(db):
    offs = db.offsets.astype(np.int64)
    return [(int(offs[k]), int(offs[k + 1])) for k in range(db.num_keys)]


def _chunks(values, b, e, top):
    """f32 score[cnt] then u16 cell[cnt] per chunk of <= 64 postings, cell = top - branch."""
    out = bytearray()
    for c0 in range(b, e, 64):
        part = values[c0:min(e, c0 + 64)]
        out += part["score"].astype("<f4").tobytes()
        out += (top - part["branch"].astype(np.int64)).astype("<u2").tobytes()
    return bytes(out)


def _pad(b, n):
    return b + bytes(-len(b) % n)


def _is_run(values, b, e):
    """A list whose branches are one ascending run b, b + 1, ...: stored without its cells (place_device.hpp, kRuns)."""
    br = values["branch"][b:e].astype(np.int64)
    return e > b and bool((np.diff(br) == 1).all())


def _reference_packed(db, n_pad, runs=True):
    """lens[key] = the first word of the table entry: len | first cell << 16 for a run (scores only in the posting
    region, 4 bytes each), len alone for a list with explicit cells (6 bytes per posting)."""
    top = n_pad - 1
    post, lens, lines = bytearray(), [], []
    for b, e in _lists(db):
        run = runs and _is_run(db.values, b, e)
        lens.append((e - b) | ((top - int(db.values["branch"][b])) << 16 if run else 0))
        lines.append(len(post) // LINE)
        post += _pad(db.values["score"][b:e].astype("<f4").tobytes() if run else _chunks(db.values, b, e, top), LINE)
    post += bytes(512)
    return np.array(lens, np.uint32), np.array(lines, np.uint32), bytes(post)


@pytest.fixture(scope="module")
def scattered_db():
    tree = synth.make_tree(60, seed=7)
    return synth.make_db(tree.num_nodes, kmer_size=5, seed=8, p_present=0.5, lognormal=(2.5, 1.5), scattered=True)


@pytest.fixture(scope="module")
def db():
    tree = synth.make_tree(60, seed=7)                       # N = 119
    return synth.make_db(tree.num_nodes, kmer_size=5, seed=8, p_present=0.5, lognormal=(2.5, 1.5))


@pytest.fixture(scope="module")
def amino_db():
    tree = synth.make_tree(20, seed=9)
    return synth.make_db(tree.num_nodes, states="amino", kmer_size=3, seed=10, p_present=0.1, lognormal=(1.0, 1.0))


@pytest.mark.parametrize("which,runs", [("runs", True), ("scattered", True), ("runs", False)])
def test_packed_and_paired_tables(db, scattered_db, which, runs, monkeypatch):
    """`runs`: every list of the database a contiguous run (the SURVEY 8d model) -- stored without cells;
    `scattered`: lists of arbitrary distinct branches -- a few happen to be runs, most keep their cells;
    EPIK_AMD_RUNS=0: every list explicit."""
    db = db if which == "runs" else scattered_db
    # (by itself the builder run-codes databases of more than 512 MB only: DESIGN.md 4)
    monkeypatch.setenv("EPIK_AMD_RUNS", "1" if runs else "0")
    n_pad = (db.num_branches + 1 + 63) // 64 * 64
    lens, lines, post = _reference_packed(db, n_pad, runs)
    if runs:
        n_runs = int(((lens >> 16) != 0).sum())
        assert (n_runs > 0.9 * int((lens != 0).sum())) if which == "runs" else (0 < n_runs < 0.5 * int((lens != 0).sum()))
    monkeypatch.setenv("EPIK_AMD_KERNEL", "wave")
    monkeypatch.setenv("EPIK_AMD_LAYOUT", "packed")
    plan, table, filt, postings = eplacer.build_image(db)
    assert plan.kernel == 0 and plan.layout == 2 and plan.filter_bytes == 0
    assert postings.tobytes() == post
    assert table[:-8].view(np.uint32).reshape(-1, 2).tolist() == np.stack([lens, lines], 1).tolist()
    # paired: block X = entries of a.X (slots 0-3) and X.b (slots 4-7), every code twice
    monkeypatch.setenv("EPIK_AMD_LAYOUT", "paired")
    plan, table, filt, postings = eplacer.build_image(db)
    assert plan.layout == 3 and postings.tobytes() == post
    blocks = table[:-8].view(np.uint32).reshape(-1, 8, 2)
    q = db.num_keys // 4
    for x in range(q):
        for a in range(4):
            assert blocks[x, a].tolist() == [lens[a * q + x], lines[a * q + x]]
            assert blocks[x, 4 + a].tolist() == [lens[x * 4 + a], lines[x * 4 + a]]


@pytest.mark.parametrize("records", ["wide", "narrow"])
def test_filtered_layout_and_shard(amino_db, monkeypatch, records):
    """records: the presence filter in 64-bit words, or its 2 x 20 bits packed into 5 bytes (EPIK_AMD_FILTER)."""
    db = amino_db
    monkeypatch.setenv("EPIK_AMD_KERNEL", "wave")
    monkeypatch.setenv("EPIK_AMD_LAYOUT", "filtered")
    monkeypatch.setenv("EPIK_AMD_FILTER", records)
    sigma, blocks = 20, db.num_keys // 20
    for shard_index, shard_count in [(0, 1), (1, 3)]:
        plan, table, filt, postings = eplacer.build_image(db, shard_index=shard_index, shard_count=shard_count)
        assert plan.layout == 4
        lens = np.diff(db.offsets.astype(np.int64))
        lens[np.arange(db.num_keys) % shard_count != shard_index] = 0
        assert plan.kept_entries == int(lens.sum())
        assert (table[:-8].view(np.uint32).reshape(-1, 2)[:, 0] & 0xFFFF).tolist() == lens.tolist()
        if records == "narrow":
            assert len(filt) == blocks * 5 + 8 and not filt[blocks * 5:].any()
            packed = np.zeros((blocks, 8), dtype=np.uint8)
            packed[:, :5] = filt[:blocks * 5].reshape(blocks, 5)
            words = packed.view(np.uint64).reshape(-1)
        else:
            words = filt.view(np.uint64)
        for x in range(blocks):
            want = 0
            for a in range(sigma):
                want |= int(lens[a * blocks + x] != 0) << a
                want |= int(lens[x * sigma + a] != 0) << (sigma + a)
            assert int(words[x]) == want


def test_compact_layout(db, monkeypatch):
    monkeypatch.setenv("EPIK_AMD_KERNEL", "wave")
    monkeypatch.setenv("EPIK_AMD_LAYOUT", "compact")
    n_pad = (db.num_branches + 1 + 63) // 64 * 64
    plan, table, _, postings = eplacer.build_image(db, shard_index=1, shard_count=2)
    lens = np.diff(db.offsets.astype(np.int64))
    lens[np.arange(db.num_keys) % 2 != 1] = 0
    assert table.view(np.uint64).tolist() == np.concatenate([[0], np.cumsum(lens)]).tolist()
    rec = postings[:-512].view(np.dtype([("score", "<f4"), ("cell", "<u4")]))
    keep = np.repeat(lens != 0, np.diff(db.offsets.astype(np.int64)))
    assert rec["score"].tobytes() == db.values["score"][keep].tobytes()
    assert rec["cell"].tolist() == (n_pad - 1 - db.values["branch"][keep].astype(np.int64)).tolist()


@pytest.mark.parametrize("kernel,waves,table", [("team4", 4, "paired"), ("team4", 4, "plain"), ("team8", 8, "plain"),
                                                ("team2", 2, "paired"), ("team2", 2, "plain")])
def test_team_layout(db, kernel, waves, table, monkeypatch):
    """Every list as W sublists, one per slice of the branch range, in the list's order; cells local to
    the slice; one {line, len[W]} entry per code (8 bytes with two slices per pass, else 16 / 32) -- with 4 letters and
    entries of up to 16 bytes twice, in the block
    of the (k-1)-mer it ends with (slot = its first letter) and in the block of the (k-1)-mer it starts
    with (slot = 4 + its last letter)."""
    monkeypatch.setenv("EPIK_AMD_KERNEL", kernel)
    if table == "plain":
        monkeypatch.setenv("EPIK_AMD_TEAM_TABLE", "plain")
    plan, table_bytes, _, postings = eplacer.build_image(db)
    assert plan.kernel == 1 and plan.layout == 5 and plan.team_waves == waves and plan.team_passes == 1
    rows = plan.slice_rows
    assert rows * waves >= db.num_branches > rows * (waves - 1)
    rows_pad = (rows + 1 + 15) // 16 * 16
    entry_bytes = {2: 8, 4: 16, 8: 32}[waves]
    paired = table == "paired"
    assert len(table_bytes) == db.num_keys * entry_bytes * (2 if paired else 1)
    blocks = db.num_keys // 4

    def entries_of(key):
        if not paired:
            return [table_bytes[key * entry_bytes:(key + 1) * entry_bytes]]
        as_suffix = ((key % blocks) * 8 + key // blocks) * entry_bytes       # a.X in block X, slot a
        as_prefix = ((key // 4) * 8 + 4 + key % 4) * entry_bytes            # X.b in block X, slot 4 + b
        return [table_bytes[as_suffix:as_suffix + entry_bytes], table_bytes[as_prefix:as_prefix + entry_bytes]]

    post, line = bytearray(), 0
    for key, (b, e) in enumerate(_lists(db)):
        v = db.values[b:e]
        region = bytearray()
        lens = []
        for w in range(waves):
            sub = v[(v["branch"] // rows) == w].copy()
            lens.append(len(sub))
            sub["branch"] -= w * rows
            region += _pad(_chunks(sub, 0, len(sub), rows_pad - 1), 4)
        for entry in entries_of(key):
            assert entry[4:4 + 2 * waves].view(np.uint16).tolist() == lens
            if e > b:
                assert int(entry[:4].view(np.uint32)[0]) == line
        region = _pad(bytes(region), LINE)
        post += region
        line += len(region) // LINE
    assert postings.tobytes() == bytes(post) + bytes(512)


@pytest.mark.parametrize("how", ["create_sharded", "descriptor"])
@pytest.mark.parametrize("kernel,waves", [("team4", 4), ("team2", 2), ("team8", 8)])
def test_team_layout_of_a_shard(db, kernel, waves, how, monkeypatch):
    """A k-mer-space shard (g of G) of the sliced layout knows that it is one: its table holds an entry per code OF THE
    SHARD, code // G being the entry's place -- 1 / G of the entries, never paired -- and the postings are those of
    the shard's lists alone; the same image whether create_sharded() picks the shard out of a whole database or the
    descriptor holds the shard already and says so (desc.shard)."""
    import copy
    monkeypatch.setenv("EPIK_AMD_KERNEL", kernel)
    g, G = 1, 3
    whole = eplacer.plan(db)
    if how == "create_sharded":
        plan, table_bytes, _, postings = eplacer.build_image(db, shard_index=g, shard_count=G)
    else:
        lens = np.diff(db.offsets.astype(np.int64))
        mine = np.arange(db.num_keys) % G == g
        sub = copy.copy(db)
        sub.offsets = np.concatenate([[0], np.cumsum(np.where(mine, lens, 0))]).astype(np.uint64)
        sub.values = db.values[np.repeat(mine, lens)].copy()
        sub.shard = (g, G)
        plan, table_bytes, _, postings = eplacer.build_image(sub)
        sub.shard = (0, G)   # the descriptor says shard 0 and holds shard 1's lists: refused
        with pytest.raises(capi.EpikAmdError, match="another shard"):
            eplacer.plan(sub)
    assert plan.kernel == 1 and plan.layout == 5 and plan.team_waves == waves
    rows = plan.slice_rows
    rows_pad = (rows + 1 + 15) // 16 * 16
    entry_bytes = {2: 8, 4: 16, 8: 32}[waves]
    n_mine = len(range(g, db.num_keys, G))
    assert len(table_bytes) == n_mine * entry_bytes
    # (the whole database's table: every code, twice where it is paired)
    assert whole.table_bytes == db.num_keys * entry_bytes * (2 if waves <= 4 else 1)
    post, line = bytearray(), 0
    for key, (b, e) in enumerate(_lists(db)):
        if key % G != g:
            continue
        v = db.values[b:e]
        region = bytearray()
        lens = []
        for w in range(waves):
            sub_list = v[(v["branch"] // rows) == w].copy()
            lens.append(len(sub_list))
            sub_list["branch"] -= w * rows
            region += _pad(_chunks(sub_list, 0, len(sub_list), rows_pad - 1), 4)
        entry = table_bytes[(key // G) * entry_bytes:(key // G + 1) * entry_bytes]
        assert entry[4:4 + 2 * waves].view(np.uint16).tolist() == lens
        if e > b:
            assert int(entry[:4].view(np.uint32)[0]) == line
        region = _pad(bytes(region), LINE)
        post += region
        line += len(region) // LINE
    assert postings.tobytes() == bytes(post) + bytes(512)


def test_plan_chooses_the_kernel_by_tree_size(monkeypatch):
    monkeypatch.delenv("EPIK_AMD_KERNEL", raising=False)
    monkeypatch.delenv("EPIK_AMD_LAYOUT", raising=False)
    monkeypatch.delenv("EPIK_AMD_RUNS", raising=False)
    small = synth.make_db(999, kmer_size=6, seed=1)
    p = eplacer.plan(small)
    assert p.kernel == 0 and p.layout == 3 and list(p.resident_waves) == [20, 20, 16]
    assert p.run_coded == 0      # a database the Infinity Cache holds keeps its cells (the kernel without the run path)
    monkeypatch.setenv("EPIK_AMD_RUNS", "1")
    q = eplacer.plan(small)
    assert q.run_coded == 1 and q.posting_bytes < 0.8 * p.posting_bytes
    monkeypatch.delenv("EPIK_AMD_RUNS")
    large = synth.make_db(9999, kmer_size=6, seed=1)
    p = eplacer.plan(large)
    assert p.kernel == 1 and p.team_passes == 1 and p.team_waves * p.slice_rows >= 9999
    assert list(p.resident_waves) == [3, 2, 1]    # what one wavefront per read would get
    huge = synth.make_db(120_000, kmer_size=4, seed=1, lognormal=(6.0, 1.0))
    p = eplacer.plan(huge)                       # beyond any single-pass geometry: several passes, no bound
    assert p.kernel == 1 and p.team_passes > 1 and p.team_waves * p.team_passes * p.slice_rows >= 120_000


def test_build_streams_without_a_host_copy():
    """create() must not hold a second copy of the database on the host: the builder's own memory is
    bounded by its staging, not by the image (here 187 MB of postings, run-coded since round 5 -- the headline
    database's image then lies inside the Infinity Cache --, + 16 MB of table)."""
    tree = synth.make_tree(500, seed=42)
    db = synth.make_db(tree.num_nodes, kmer_size=10, seed=43)
    proc = psutil.Process()
    peak, stop = [0], threading.Event()

    def sample():
        while not stop.is_set():
            peak[0] = max(peak[0], proc.memory_info().rss)
            time.sleep(0.002)

    before = proc.memory_info().rss
    t = threading.Thread(target=sample)
    t.start()
    try:
        plan = eplacer.build_image(db, discard=True)[0]
    finally:
        stop.set()
        t.join()
    assert plan.posting_bytes > 180 << 20 and plan.run_coded == 1
    grown = peak[0] - before
    assert grown < 32 << 20, f"the image build grew the process by {grown >> 20} MiB"


@pytest.mark.parametrize("kernel,layout,shard", [
    ("wave", "packed", (0, 1)), ("wave", "paired", (1, 3)), ("wave", "compact", (0, 1)), ("wave", "compact", (2, 3)),
    ("team4", None, (0, 1)), ("team8x2", None, (1, 2)), ("team4x3", None, (0, 1))])
def test_sparse_descriptor_builds_the_same_image(db, kernel, layout, shard, monkeypatch):
    """ABI 3: keys[num_present] + offsets[num_present + 1] instead of an offset per possible k-mer -- the same
    device image byte for byte, whole or as a k-mer-space shard."""
    monkeypatch.setenv("EPIK_AMD_KERNEL", kernel)
    if layout:
        monkeypatch.setenv("EPIK_AMD_LAYOUT", layout)
    else:
        monkeypatch.delenv("EPIK_AMD_LAYOUT", raising=False)
    dense = eplacer.build_image(db, shard_index=shard[0], shard_count=shard[1])
    sparse = eplacer.build_image(db, shard_index=shard[0], shard_count=shard[1], sparse=True)
    for name in ("kernel", "layout", "table_bytes", "filter_bytes", "posting_bytes", "kept_entries"):
        assert getattr(dense[0], name) == getattr(sparse[0], name), name
    for a, b in zip(dense[1:], sparse[1:]):
        assert a.tobytes() == b.tobytes()


def test_sparse_descriptor_filtered_layout(amino_db, monkeypatch):
    monkeypatch.setenv("EPIK_AMD_KERNEL", "wave")
    monkeypatch.setenv("EPIK_AMD_LAYOUT", "filtered")
    for shard in ((0, 1), (1, 2)):
        dense = eplacer.build_image(amino_db, shard_index=shard[0], shard_count=shard[1])
        sparse = eplacer.build_image(amino_db, shard_index=shard[0], shard_count=shard[1], sparse=True)
        assert dense[0].layout == 4 and dense[0].filter_bytes == sparse[0].filter_bytes
        for a, b in zip(dense[1:], sparse[1:]):
            assert a.tobytes() == b.tobytes()


def test_sparse_descriptor_is_validated(db):
    import ctypes
    from epik_amd import capi
    lens = np.diff(db.offsets.astype(np.int64))
    keys = np.nonzero(lens)[0].astype(np.uint32)
    offsets = np.concatenate([[0], np.cumsum(lens[keys])]).astype(np.uint64)
    common = dict(states=db.states, kmer_size=db.kmer_size, num_branches=db.num_branches, threshold=db.threshold)

    def plan_rc(k, o):
        desc, keep = eplacer.make_desc(o, db.values, keys=k, **common)
        rc = capi.load().epik_amd_placer_plan(ctypes.byref(desc), 0, 1, 1 << 34, ctypes.byref(capi.Plan()))
        del keep
        return rc

    assert plan_rc(keys, offsets) == capi.OK
    swapped = keys.copy()
    swapped[[3, 4]] = swapped[[4, 3]]
    assert plan_rc(swapped, offsets) == capi.ERR_INVALID          # not ascending
    twice = keys.copy()
    twice[5] = twice[4]
    assert plan_rc(twice, offsets) == capi.ERR_INVALID            # a code twice
    outside = keys.copy()
    outside[-1] = db.num_keys
    assert plan_rc(outside, offsets) == capi.ERR_INVALID          # outside the key space
    bent = offsets.copy()
    bent[2] = bent[4] + 1
    assert plan_rc(keys, bent) == capi.ERR_INVALID                # offsets not monotone


def _plan_fields(p):
    return dict(kernel=p.kernel, layout=p.layout, team_waves=p.team_waves, team_passes=p.team_passes, slice_rows=p.slice_rows,
                resident_waves=list(p.resident_waves), table_bytes=p.table_bytes, filter_bytes=p.filter_bytes,
                kept_entries=p.kept_entries, run_coded=p.run_coded)


@pytest.mark.parametrize("case", ["nucl-wave", "nucl-wave-runs", "amino-filtered", "nucl-team4", "nucl-team2", "compact"])
@pytest.mark.parametrize("shard", [(0, 1), (1, 3)])
def test_plan_from_the_sizes_alone_is_the_plan_of_the_database(case, shard, monkeypatch):
    """epik_amd_placer_plan_sizes -- the tree, the key space and a histogram of list lengths, no posting -- gives the
    plan epik_amd_placer_plan gives on the database itself: kernel, layout, geometry, table and filter always; the
    posting region exactly for the layouts of the one-wavefront kernel (run-coded or not) and as an upper bound,
    within a few percent, for the sliced layout (how a list falls over the slices is not in a histogram)."""
    for var in ("EPIK_AMD_KERNEL", "EPIK_AMD_LAYOUT", "EPIK_AMD_RUNS"):
        monkeypatch.delenv(var, raising=False)
    free = 288 << 30
    if case == "amino-filtered":
        db = synth.make_db(120, states="amino", kmer_size=3, p_present=0.2, seed=9, lognormal=(2.0, 1.0))
    elif case in ("nucl-team4", "nucl-team2"):
        leaves = 5000 if case == "nucl-team4" else 1200
        db = synth.make_db(2 * leaves - 1, kmer_size=7, p_present=0.6, seed=7)
    else:
        db = synth.make_db(999, kmer_size=7, p_present=0.6, seed=11)
        if case == "nucl-wave-runs":
            monkeypatch.setenv("EPIK_AMD_RUNS", "1")
        if case == "compact":
            free = 1 << 19   # a device whose free memory the table would not fit a quarter of
            # (the compact layout keeps the caller's offset width; plan_sizes assumes the narrowest that holds the postings)
            db.offsets = db.offsets.astype(np.uint32)
    g, G = shard
    real = eplacer.plan(db, shard_index=g, shard_count=G, free_bytes=free)
    bins = eplacer.list_bins(db, g, G)
    sized = eplacer.plan_sizes(states=db.states, kmer_size=db.kmer_size, num_branches=db.num_branches, bins=bins,
                               shard_index=g, shard_count=G, free_bytes=free)
    assert _plan_fields(sized) == _plan_fields(real)
    if real.kernel == 1:
        assert sized.posting_bytes_is_bound == 1
        assert real.posting_bytes <= sized.posting_bytes <= real.posting_bytes * 1.08
        if G > 1:   # a shard's table: an entry per code of the shard, unpaired
            assert real.table_bytes == len(range(g, 4 ** db.kmer_size, G)) * (8 if real.team_waves == 2 else 16)
    else:
        assert sized.posting_bytes_is_bound == 0 and sized.posting_bytes == real.posting_bytes
    if case == "nucl-wave-runs":
        assert real.run_coded == 1
    if case == "compact":
        assert real.layout in (0, 1)


def test_a_database_beyond_one_gpu_plans_as_shards_that_fit():
    """BASELINE configs[4] in numbers only: N = 9 999, nucl k = 14 (268 M codes), 60 G postings -- 0.37 TB of posting
    region, more than one MI355X holds -- planned without a posting: the whole database does not fit 288 GB, two shards
    do, and each of eight takes an eighth of the table as well (a shard knows its modulus: an entry per code of its own)."""
    hbm = 288 << 30
    codes = 4 ** 14
    present = int(codes * 0.93)
    bins = [(1, present // 4, present // 4), (40, present // 4, present // 8), (200, present // 4, present // 4), (655, present - 3 * (present // 4), 0)]
    total = sum(b[0] * b[1] for b in bins)
    assert total > 55e9
    common = dict(states="nucl", kmer_size=14, num_branches=9999)
    whole = eplacer.plan_sizes(bins=bins, **common)
    assert whole.kernel == 1 and whole.team_waves == 4 and whole.kept_entries == total
    assert whole.table_bytes == codes * 32 and whole.table_bytes + whole.posting_bytes > hbm
    for G in (2, 8):
        for g in (0, G - 1):
            # (a shard's lists: every G-th code -- the lengths spread evenly over the residues)
            mine = [(b[0], b[1] // G, b[2] // G) for b in bins]
            part = eplacer.plan_sizes(bins=mine, shard_index=g, shard_count=G, **common)
            assert part.table_bytes == (codes // G) * 16
            assert part.table_bytes + part.posting_bytes < hbm * 0.8
            assert part.kept_entries * G <= total
    with pytest.raises(capi.EpikAmdError, match="more lists than the shard has codes"):
        eplacer.plan_sizes(bins=bins, shard_index=0, shard_count=8, **common)
