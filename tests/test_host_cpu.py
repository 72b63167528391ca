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
    assert mu.returncode == 255 and "Mu has to" in mu.stderr


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
    # legacy semantics: same edge set counts as a match even with different likelihoods
    leg = json.loads(json.dumps(parsed))
    leg["q1"][0]["likelihood"] = -1.0
    assert jplace_diff.diff_legacy(parsed, leg) == []
