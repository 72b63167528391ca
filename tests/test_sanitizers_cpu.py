"""Sanitizer runs of the CPU side: the host unit tests under ASan + UBSan, the driver's
reader / device threads / ordered writer pipeline (epik_amd/host/main.cpp) under ThreadSanitizer and
under ASan + UBSan against a test-only stub of the C ABI (tests/stub/epik_amd_stub.c, canned rows --
no GPU here), and the oracle under ASan + UBSan with four OpenMP threads."""
import json
import os
import subprocess

import numpy as np
import pytest

from epik_amd import dbfile, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "epik_amd", "bin", "san")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
           TSAN_OPTIONS="halt_on_error=1")


@pytest.fixture(scope="module")
def san_bins():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "epik_amd", "host"), "sanitize"], capture_output=True,
                         text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return SAN


def _clean(run):
    text = run.stdout + run.stderr
    assert run.returncode == 0, text[-3000:]
    for needle in ("ERROR: AddressSanitizer", "WARNING: ThreadSanitizer", "runtime error:", "LeakSanitizer"):
        assert needle not in text, text[-3000:]


def test_host_units_under_asan_ubsan(san_bins, tmp_path):
    run = subprocess.run([os.path.join(san_bins, "host_test_asan"), str(tmp_path)], capture_output=True, text=True,
                         env=ENV, timeout=300)
    _clean(run)
    assert "host tests ok" in run.stdout


@pytest.fixture(scope="module")
def driver_case(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("san")
    tree = synth.make_tree(30, seed=5)
    db = synth.make_db(tree.num_nodes, kmer_size=5, seed=6, p_present=0.5)
    db_path = str(tmp / "db.ekdb")
    dbfile.write_db(db_path, db, tree.newick())
    rng = np.random.default_rng(7)
    fasta = str(tmp / "q.fasta")
    n = 30_000
    with open(fasta, "w") as fh:
        for i in range(n):
            seq = "".join(rng.choice(list("ACGT"), size=int(rng.integers(3, 120))))
            fh.write(f">read {i}\n{seq}\n")
            if i % 97 == 0:
                fh.write(f">twin {i}\n{seq}\n")
    return tmp, db_path, fasta, n + (n + 96) // 97


@pytest.mark.parametrize("binary,devices,jobs,shards", [
    ("epik-dna_tsan", "0,1", "4", "1"), ("epik-dna_tsan", "0", "1", "1"), ("epik-dna_asan", "0,1", "3", "1"),
    ("epik-dna_tsan", "0,1", "2", "2"), ("epik-dna_asan", "0,1", "2", "3")])   # --db-shard: shards loaded one by one
def test_driver_pipeline_under_sanitizers(san_bins, driver_case, binary, devices, jobs, shards):
    """Small batches, two device threads, several formatting threads: every hand-over of the pipeline
    is exercised many times; the output must name every read once, in input order."""
    tmp, db_path, fasta, n_records = driver_case
    out_dir = tmp / f"out_{binary}_{devices.replace(',', '_')}_{shards}"
    out_dir.mkdir()
    run = subprocess.run([os.path.join(san_bins, binary), "-d", db_path, "-q", fasta, "-o", str(out_dir), "--devices",
                          devices, "--batch-size", "500", "-j", jobs, "--db-shard", shards], capture_output=True,
                         text=True, env=ENV, timeout=600)
    _clean(run)
    assert f"Placed {n_records} sequences." in run.stdout
    with open(out_dir / "placements_q.fasta.jplace") as fh:
        doc = json.load(fh)
    names = [nm[0] for obj in doc["placements"] for nm in obj["nm"]]
    assert len(names) == n_records and len(set(names)) == n_records
    firsts = [int(obj["nm"][0][0].split()[1]) for obj in doc["placements"]]
    assert firsts == sorted(firsts), "batches must be written in input order"


def test_oracle_under_asan_ubsan():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    run = subprocess.run([os.path.join(ROOT, "oracle", "oracle_san_test")], capture_output=True, text=True,
                         env=dict(ENV, OMP_NUM_THREADS="4"), timeout=300)
    _clean(run)
    assert "oracle sanitizer run ok" in run.stdout
