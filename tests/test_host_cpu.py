"""CPU tests of the callers on either side of the path: FASTA batches, Newick, jplace
writer, --max-ram parsing, the EPIKAMD1 container (C++ unit binary + Python twins), the
epik.py launcher and the jplace differ."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from epik_amd import dbfile, jplace, jplace_diff, synth
from epik_amd.placer import PlacedCollection, PlacedSequence, Placement, pendant_lengths

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "epik_amd", "bin")


@pytest.fixture(scope="module")
def host_bins():
    subprocess.run(["make", "-C", os.path.join(ROOT, "epik_amd", "csrc")], check=True,
                   stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(ROOT, "epik_amd", "host")], check=True,
                   stdout=subprocess.DEVNULL)
    return BIN


def test_cpp_host_units(host_bins, tmp_path):
    out = subprocess.run([os.path.join(host_bins, "host_test"), str(tmp_path)], capture_output=True,
                         text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host tests ok" in out.stdout


def test_driver_help_and_errors(host_bins, tmp_path):
    """CI of the reference checks only this much (build.yml:61-66); plus the exit codes of
    main.cpp:272-283,384-388."""
    for name in ("epik-dna", "epik-aa"):
        out = subprocess.run([os.path.join(host_bins, name), "--help"], capture_output=True, text=True)
        assert out.returncode == 0 and "--keep-at-most" in out.stdout and "--max-ram" in out.stdout
    bad = subprocess.run([os.path.join(host_bins, "epik-dna"), "-d", "/nonexistent", "-q", "x", "-o", "."],
                         capture_output=True, text=True)
    assert bad.returncode == 255 and "Error:" in bad.stderr   # return -1
    mu = subprocess.run([os.path.join(host_bins, "epik-dna"), "-d", "x", "-q", "x", "-o", ".", "--mu", "2"],
                        capture_output=True, text=True)
    assert mu.returncode == 255 and "--mu must lie in [0, 1]" in mu.stderr


def test_launcher_builds_reference_argv():
    sys.path.insert(0, ROOT)
    import epik
    argv = epik.driver_command(database="db.ekdb", states="amino", omega=1.5, mu=0.5, outputdir="out",
                               threads=4, max_ram="4G", gpus=8, input_file="q.fasta")
    assert argv[0].endswith("epik-aa")
    # reference epik.py:85-96: -d DB -q IN -j T --omega W --mu M -o OUT [--max-ram R] IN
    assert argv[1:13] == ["-d", "db.ekdb", "-q", "q.fasta", "-j", "4", "--omega", "1.5", "--mu", "0.5",
                          "-o", "out"]
    assert argv[13:] == ["--max-ram", "4G", "--gpus", "8", "q.fasta"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "epik.py"), "place", "--help"],
                         capture_output=True, text=True)
    assert out.returncode == 0
    for flag in ("--database", "--states", "--omega", "--mu", "--outputdir", "--threads", "--max-ram"):
        assert flag in out.stdout


@pytest.mark.skipif(not os.path.exists("/root/reference/epik.py"), reason="the reference is not present on this machine")
@pytest.mark.parametrize("states,max_ram", [("nucl", None), ("amino", "4G")])
def test_launcher_argv_equals_the_reference_launcher(monkeypatch, capsys, states, max_ram):
    """The reference's own `place_queries` (epik.py:73-98) with subprocess.call intercepted,
    beside ours: same flags, same order, same values; ours only picks its own binary."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("reference_epik", "/root/reference/epik.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    seen = []
    monkeypatch.setattr(ref.subprocess, "call", lambda command: seen.append(list(command)) or 0)
    ref.place_queries("db.ipk", states, 1.5, 0.8, "out", 3, max_ram, "q.fasta")
    capsys.readouterr()
    sys.path.insert(0, ROOT)
    import epik
    ours = epik.driver_command(database="db.ipk", states=states, omega=1.5, mu=0.8, outputdir="out", threads=3,
                               max_ram=max_ram, gpus=1, input_file="q.fasta")   # one GPU: no extra flag
    assert os.path.basename(ours[0]) == os.path.basename(seen[0][0])      # epik-dna / epik-aa
    assert ours[1:] == seen[0][1:]


def test_dbfile_roundtrip_and_filters(tmp_path):
    tree = synth.make_tree(8, seed=1)
    db = synth.make_db(tree.num_nodes, kmer_size=4, p_present=0.7, seed=5, lognormal=(1.0, 1.0))
    path = str(tmp_path / "db.ekdb")
    dbfile.write_db(path, db, tree.newick())
    back, newick = dbfile.read_db(path, mu=1.0, omega=1.5)
    assert newick == tree.newick()
    assert np.array_equal(back.offsets, db.offsets) and np.array_equal(back.values, db.values)
    half, _ = dbfile.read_db(path, mu=0.5, omega=1.5)
    assert 0 < half.num_entries < db.num_entries
    capped, _ = dbfile.read_db(path, mu=1.0, omega=1.5, max_entries=100)
    assert 0 < capped.num_entries <= 100
    stricter, _ = dbfile.read_db(path, mu=1.0, omega=2.0)
    assert stricter.num_entries < db.num_entries
    assert (stricter.values["score"] >= float(stricter.log_threshold)).all()


def test_loader_refuses_damaged_and_foreign_containers(host_bins, tmp_path):
    """main.cpp:277-283 loads and checks the version; here the container has a trailer (record count, posting count,
    crc32 of the records: a conversion that died half way is caught at load), and a file that is not this build's
    container is answered with the converter and its build line."""
    tree = synth.make_tree(8, seed=1)
    db = synth.make_db(tree.num_nodes, kmer_size=4, p_present=0.7, seed=5, lognormal=(1.0, 1.0))
    good = tmp_path / "db.ekdb"
    dbfile.write_db(str(good), db, tree.newick())
    raw = good.read_bytes()
    assert raw[-dbfile.TRAILER_BYTES:][:8] == dbfile.END_MAGIC

    def driver(path):   # epik_amd::load as the driver calls it (the driver itself asks for its devices first)
        return subprocess.run([os.path.join(host_bins, "host_test"), "load", str(path)], capture_output=True, text=True)

    cut = tmp_path / "cut.ekdb"
    cut.write_bytes(raw[:-40])
    out = driver(cut)
    assert out.returncode == 255 and "truncated or was not finished" in out.stderr, out.stderr
    with pytest.raises(RuntimeError, match="trailer is missing"):
        dbfile.read_db(str(cut))

    flipped = bytearray(raw)
    flipped[len(raw) - dbfile.TRAILER_BYTES - 5] ^= 0x40    # inside the last record
    bad = tmp_path / "bad.ekdb"
    bad.write_bytes(bytes(flipped))
    out = driver(bad)
    assert out.returncode == 255 and "do not match their checksum" in out.stderr, out.stderr
    with pytest.raises(RuntimeError, match="checksum"):
        dbfile.read_db(str(bad))

    fewer = bytearray(raw)
    fewer[-24:-16] = (int.from_bytes(raw[-24:-16], "little") - 1).to_bytes(8, "little")   # the trailer's k-mer count
    short = tmp_path / "short.ekdb"
    short.write_bytes(bytes(fewer))
    out = driver(short)
    assert out.returncode == 255 and "trailer counts" in out.stderr, out.stderr

    ipk = tmp_path / "db.ipk"
    ipk.write_bytes(b"22 serialization::archive 17 0 0" + b"\0" * 64)
    out = driver(ipk)
    assert out.returncode == 255, out.stderr
    for needle in ("not an EPIKAMD1 file", "tools/ipk2ekdb.cpp", "g++ -std=c++17", "-li2l_dna", "./ipk2ekdb"):
        assert needle in out.stderr, (needle, out.stderr)

    # a version-1 file (no trailer) still loads: the good file with its version set back and its trailer cut off
    v1 = bytearray(raw[:-dbfile.TRAILER_BYTES])
    v1[8:12] = (1).to_bytes(4, "little")
    old = tmp_path / "v1.ekdb"
    old.write_bytes(bytes(v1))
    back, _ = dbfile.read_db(str(old))
    assert np.array_equal(back.values, db.values)
    out = driver(old)
    assert out.returncode == 0 and "version 1" in out.stdout, out.stderr
    assert driver(good).returncode == 0


def test_max_ram_cuts_every_shard_at_the_same_place(host_bins, tmp_path):
    """--max-ram with --db-shard: the limit is per shard, the cut ONE position of the file -- the union of the shards
    is the prefix an unsharded load with the summed limit keeps (advisor, round 3)."""
    tree = synth.make_tree(8, seed=1)
    db = synth.make_db(tree.num_nodes, kmer_size=4, p_present=0.7, seed=5, lognormal=(1.0, 1.0))
    path = str(tmp_path / "db.ekdb")
    dbfile.write_db(path, db, tree.newick())

    def keys(limit, index, count):
        out = subprocess.run([os.path.join(host_bins, "host_test"), "load", path, str(limit), str(index), str(count)],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        line = [x for x in out.stdout.splitlines() if x.startswith("keys")][0]
        return [int(x) for x in line.split()[1:]]

    per_shard, shards = 60, 3
    whole = keys(per_shard * shards, 0, 1)
    parts = [keys(per_shard, g, shards) for g in range(shards)]
    assert 0 < len(whole) < int((np.diff(db.offsets.astype(np.int64)) != 0).sum())      # the limit really cuts
    assert sorted(k for p in parts for k in p) == whole
    assert all(k % shards == g for g, p in enumerate(parts) for k in p)


def test_pendant_lengths_formula():
    """place.cpp:110-123: distal = len/2; pendant = subtree mean (if > 1 node) + distal."""
    distal, pendant = pendant_lengths(np.array([0.2, 0.4, 1.0]), np.array([1, 1, 3]),
                                      np.array([0.0, 0.0, 0.6]))
    assert list(distal) == [0.1, 0.2, 0.5]
    assert pendant[0] == 0.1 and pendant[2] == pytest.approx(0.6 / 3 + 0.5)


def _collection():
    rows = [Placement(3, -1.5, 0.75, 2, 0.05, 0.15), Placement(1, -2.0, 0.25, 1, 0.1, 0.2)]
    return PlacedCollection(sequence_map={"ACGT": ["q1", "q2"], "TT": ["q3"]},
                            placed_seqs=[PlacedSequence("ACGT", rows), PlacedSequence("TT", [])])


def test_jplace_write_read_and_diff(tmp_path):
    a = str(tmp_path / "a.jplace")
    jplace.write_jplace(a, _collection(), "epik.py place", "(A:1{0},B:2{1}):0{2};")
    doc = json.load(open(a))
    assert doc["version"] == 3 and doc["fields"] == jplace.FIELDS
    assert doc["placements"][0]["nm"] == [["q1", 1], ["q2", 1]]
    parsed = jplace.read_jplace(a)
    assert set(parsed) == {"q1", "q2", "q3"} and parsed["q2"][0]["edge_num"] == 3
    assert jplace_diff.diff_strict(parsed, parsed) == []
    # a different LWR beyond 1e-5, a swapped order, a missing name
    other = json.loads(json.dumps(parsed))
    other["q1"][0]["like_weight_ratio"] += 2e-5
    assert any("like_weight_ratio" in p for p in jplace_diff.diff_strict(parsed, other))
    other = json.loads(json.dumps(parsed))
    other["q1"] = other["q1"][::-1]
    assert any("edges" in p for p in jplace_diff.diff_strict(parsed, other))
    other = {k: v for k, v in parsed.items() if k != "q3"}
    assert any("one file only" in p for p in jplace_diff.diff_strict(parsed, other))
    # exact likelihood ties compare as sets
    tie_a = {"x": [dict(edge_num=1, likelihood=-2.0, like_weight_ratio=0.5),
                   dict(edge_num=2, likelihood=-2.0, like_weight_ratio=0.5)]}
    tie_b = {"x": tie_a["x"][::-1]}
    assert jplace_diff.diff_strict(tie_a, tie_b) == []
    # legacy semantics (the reference's script): LWR is ignored, likelihoods are compared
    leg = json.loads(json.dumps(parsed))
    leg["q1"][0]["like_weight_ratio"] = 0.123
    assert jplace_diff.diff_legacy(parsed, leg) == []
    leg["q1"][0]["likelihood"] = -1.0
    assert "q1" in jplace_diff.diff_legacy(parsed, leg)


REFERENCE_DIFFER = "/root/reference/scripts/jplace_diff.py"


@pytest.mark.skipif(not os.path.exists(REFERENCE_DIFFER), reason="the reference is not present on this machine")
def test_legacy_differ_agrees_with_the_reference_script(tmp_path, capsys):
    """`--legacy` against the reference's own comparison function (its CLI wrapper is broken, the
    function behind it is callable), on jplace pairs with every kind of disagreement."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("reference_jplace_diff", REFERENCE_DIFFER)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.default_rng(7)
    names = [f"s{i}" for i in range(60)]

    def rows(n):
        edges = rng.choice(40, size=n, replace=False)
        return [dict(edge_num=int(e), likelihood=float(-rng.uniform(0.5, 6.0)), like_weight_ratio=float(rng.random()),
                     distal_length=0.1, pendant_length=0.2) for e in edges]

    first = {name: rows(int(rng.integers(0, 6))) for name in names}
    second = json.loads(json.dumps(first))
    for i, name in enumerate(names):
        r = second[name]
        kind = i % 6
        if kind == 1 and r:
            r[0]["likelihood"] += 5e-5           # inside the 1e-4 window
        elif kind == 2 and r:
            r[0]["likelihood"] -= 0.5            # outside: compared edge by edge on 10**likelihood
        elif kind == 3 and r:
            r[0]["edge_num"] = 99                # another edge with the same likelihood: still a "match" there
        elif kind == 4 and r:
            r.pop()                              # a row missing
        elif kind == 5:
            r.append(dict(edge_num=77, likelihood=-9.0, like_weight_ratio=0.0, distal_length=0.0, pendant_length=0.0))

    def write(path, placed):
        doc = {"fields": jplace.FIELDS, "version": 3, "tree": "(A:1{0},B:2{1}):0{2};", "metadata": {},
               "placements": [{"p": [[r[f] for f in jplace.FIELDS] for r in rws], "nm": [[name, 1]]}
                              for name, rws in placed.items()]}
        with open(path, "w") as fh:
            json.dump(doc, fh)

    a, b = str(tmp_path / "a.jplace"), str(tmp_path / "b.jplace")
    write(a, first)
    write(b, second)
    for only_best in (False, True):
        capsys.readouterr()
        ref.jplace_diff.callback(a, b, only_best)
        out = capsys.readouterr().out
        matched = int(out.strip().splitlines()[-1].split("/")[0])
        ours = jplace_diff.diff_legacy(jplace.read_jplace(a), jplace.read_jplace(b), only_best)
        assert matched == len(names) - len(ours), (only_best, matched, ours)


def test_legacy_differ_matches_the_reference_verdicts_on_record():
    """The same comparison against verdicts the reference's function gave in the build container
    (tests/golden/make_jplace_diff_golden.py): this one runs wherever the reference is absent too."""
    with open(os.path.join(ROOT, "tests", "golden", "jplace_diff_reference.json")) as fh:
        golden = json.load(fh)
    assert len(golden["cases"]) == 8
    for case in golden["cases"]:
        ours = jplace_diff.diff_legacy(case["first"], case["second"], case["only_best"])
        assert case["names"] - len(ours) == case["reference_matched"], (case["seed"], case["only_best"], ours)
