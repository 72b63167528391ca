"""End to end through the native driver: EPIKAMD1 database + FASTA -> `epik.py place`
-> epik-dna -> jplace, compared (strict differ: edge order, |dLWR| <= 1e-5) with a
jplace assembled from the CPU oracle's rows."""
import os
import subprocess
import sys

import numpy as np
import pytest

from epik_amd import dbfile, jplace, jplace_diff, synth
from epik_amd.placer import PlacedCollection, PlacedSequence, Placement, pendant_lengths

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_jplace(path, oracle_lib, db, tree, records):
    seq_map = {}
    for h, s in records:
        seq_map.setdefault(s, []).append(h)
    unique = list(seq_map)
    data, offs = synth.pack_reads(unique)
    rows, n_rows, counts = oracle_lib.Oracle.from_synth(db).place(data, offs)
    distal, pendant = pendant_lengths(tree.branch_length, tree.subtree_num_nodes, tree.subtree_total_length)
    placed = []
    for i, s in enumerate(unique):
        pl = []
        for r in range(n_rows[i]):
            b, c = int(rows[i, r]["branch"]), int(counts[i, r])
            pl.append(Placement(b, float(rows[i, r]["score"]), float(rows[i, r]["lwr"]), c,
                                float(distal[b]) if c else 0.0, float(pendant[b]) if c else 0.0))
        placed.append(PlacedSequence(s, pl))
    jplace.write_jplace(path, PlacedCollection(seq_map, placed), "oracle", tree.newick(jplace=True))


@pytest.mark.parametrize("devices,kernel", [("0", None), ("0,0", None), ("0", "team4"), ("0,0", "team4x2"),
                                            ("0,0,0,0,0,0,0,0", None)])  # the eight placer threads of a node, on one device
def test_epik_py_place_matches_oracle(tmp_path, oracle_lib, devices, kernel, monkeypatch):
    """kernel: None = what create() chooses for this tree (one wavefront per read); team4 / team4x2 = the driver
    and the host-buffer entry point over the kernels of large trees (front + streaming + merge; the reads with
    an N go down the one-kernel path behind them)."""
    if kernel:
        monkeypatch.setenv("EPIK_AMD_KERNEL", kernel)
    subprocess.run(["make", "-C", os.path.join(ROOT, "epik_amd", "host")], check=True, stdout=subprocess.DEVNULL)
    tree = synth.make_tree(60, seed=11)
    db = synth.make_db(tree.num_nodes, kmer_size=8, seed=12, p_present=0.5)
    db_path = str(tmp_path / "db.ekdb")
    dbfile.write_db(db_path, db, tree.newick())
    rng = np.random.default_rng(4)
    records = []
    for i in range(5000):
        alpha = "ACGT" if i % 5 else "ACGTN"
        records.append((f"read_{i}", "".join(rng.choice(list(alpha), size=int(rng.integers(5, 250))))))
    records += [("dup_a", records[0][1]), ("dup_b", records[0][1]), ("short", "ACG")]
    fasta = str(tmp_path / "q.fasta")
    with open(fasta, "w") as fh:
        for h, s in records:
            fh.write(f">{h}\n")
            for j in range(0, len(s), 70):
                fh.write(s[j:j + 70] + "\n")
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    if devices == "0":
        cmd = [sys.executable, os.path.join(ROOT, "epik.py"), "place", "-i", db_path, "-o", str(out_dir), fasta]
    else:  # two handles on the one GPU: exercises the multi-device sharding of the driver
        cmd = [os.path.join(ROOT, "epik_amd", "bin", "epik-dna"), "-d", db_path, "-q", fasta, "-o", str(out_dir),
               "--devices", devices, "--batch-size", "777"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert "Placed 5003 sequences." in run.stdout and "Placement time:" in run.stdout
    got_path = str(out_dir / "placements_q.fasta.jplace")
    ref_path = str(tmp_path / "ref.jplace")
    _oracle_jplace(ref_path, oracle_lib, db, tree, records)
    got, ref = jplace.read_jplace(got_path), jplace.read_jplace(ref_path)
    assert set(got) == set(ref) == {h for h, _ in records}
    assert jplace_diff.diff_strict(got, ref) == []
    # lengths are joined per branch on the host (place.cpp:110-123, 435-437)
    for name in ("read_1", "dup_b"):
        for x, y in zip(got[name], ref[name]):
            assert x["distal_length"] == pytest.approx(y["distal_length"], rel=1e-9)
            assert x["pendant_length"] == pytest.approx(y["pendant_length"], rel=1e-9)
    assert got["short"] == []


def test_epik_aa_places_proteins(tmp_path, oracle_lib):
    """The protein driver (`epik.py place -s amino` -> epik-aa, epik/CMakeLists.txt:124): 20-state
    encoder, ambiguous residues B / Z / X, against the oracle's jplace."""
    from epik_amd import alphabet
    subprocess.run(["make", "-C", os.path.join(ROOT, "epik_amd", "host")], check=True, stdout=subprocess.DEVNULL)
    tree = synth.make_tree(40, seed=13)
    db = synth.make_db(tree.num_nodes, states="amino", kmer_size=4, seed=14, p_present=0.3, lognormal=(1.5, 1.0))
    db_path = str(tmp_path / "db.ekdb")
    dbfile.write_db(db_path, db, tree.newick())
    rng = np.random.default_rng(6)
    records = []
    for i in range(3000):
        alpha = alphabet.AMINO_STATES if i % 4 else alphabet.AMINO_STATES + "BZX*"
        records.append((f"prot_{i}", "".join(rng.choice(list(alpha), size=int(rng.integers(3, 320))))))
    fasta = str(tmp_path / "p.fasta")
    with open(fasta, "w") as fh:
        for h, s in records:
            fh.write(f">{h}\n{s}\n")
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    run = subprocess.run([sys.executable, os.path.join(ROOT, "epik.py"), "place", "-i", db_path, "-s", "amino",
                          "-o", str(out_dir), fasta], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert "Placed 3000 sequences." in run.stdout and "Sequence type: Proteins" in run.stdout
    ref_path = str(tmp_path / "ref.jplace")
    _oracle_jplace(ref_path, oracle_lib, db, tree, records)
    got, ref = jplace.read_jplace(str(out_dir / "placements_p.fasta.jplace")), jplace.read_jplace(ref_path)
    assert set(got) == set(ref) == {h for h, _ in records}
    assert jplace_diff.diff_strict(got, ref) == []
    # the DNA driver refuses a protein database (the two binaries differ as the reference's do)
    wrong = subprocess.run([os.path.join(ROOT, "epik_amd", "bin", "epik-dna"), "-d", db_path, "-q", fasta, "-o",
                            str(out_dir)], capture_output=True, text=True, timeout=600)
    assert wrong.returncode == 255 and "Proteins" in wrong.stderr


@pytest.mark.parametrize("shards,devices", [("2", "0,0"), ("3", "0"), ("8", "0")])  # (8: configs[4]'s eight shards, one device)
def test_db_shard_through_the_driver(tmp_path, oracle_lib, shards, devices, monkeypatch):
    """`epik.py place --db-shard G` / epik-dna --db-shard G: the database cut in G by k-mer code, every shard loaded
    on its own (the process never holds two), a handle per shard -- here on the one device --, every batch placed
    by all of them together (epik_amd_placer_place_sharded).  A tree large enough for the kernels that leave
    partial lists; reads with N follow the first-key rule over all shards.  The float32 sums of a branch are added
    shard by shard: scores agree with the oracle's to rounding, rows of equal score may swap."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "epik_amd", "host")], check=True, stdout=subprocess.DEVNULL)
    monkeypatch.setenv("EPIK_AMD_SHARD_CHUNK", "1500")   # several chunks per call: the pipelined exchange
    tree = synth.make_tree(2000, seed=15)                # N = 3 999: front + streaming + merge kernels
    db = synth.make_db(tree.num_nodes, kmer_size=8, seed=16, p_present=0.5, lognormal=(3.5, 1.5))
    db_path = str(tmp_path / "db.ekdb")
    dbfile.write_db(db_path, db, tree.newick())
    rng = np.random.default_rng(8)
    records = []
    for i in range(6000):
        alpha = "ACGT" if i % 5 else "ACGTN"
        records.append((f"read_{i}", "".join(rng.choice(list(alpha), size=int(rng.integers(5, 250))))))
    records += [("dup_a", records[0][1]), ("short", "ACG")]
    fasta = str(tmp_path / "q.fasta")
    with open(fasta, "w") as fh:
        for h, s in records:
            fh.write(f">{h}\n{s}\n")
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    if devices == "0":
        cmd = [sys.executable, os.path.join(ROOT, "epik.py"), "place", "-i", db_path, "-o", str(out_dir), "--db-shard",
               shards, fasta]
    else:
        cmd = [os.path.join(ROOT, "epik_amd", "bin", "epik-dna"), "-d", db_path, "-q", fasta, "-o", str(out_dir),
               "--devices", devices, "--db-shard", shards, "--batch-size", "777"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert "Placed 6002 sequences." in run.stdout and f"(shard 0 of {shards})" in run.stdout
    assert f"(shard {int(shards) - 1} of {shards})" in run.stdout and f"{shards} shard(s) of the database" in run.stdout
    ref_path = str(tmp_path / "ref.jplace")
    _oracle_jplace(ref_path, oracle_lib, db, tree, records)
    got, ref = jplace.read_jplace(str(out_dir / "placements_q.fasta.jplace")), jplace.read_jplace(ref_path)
    assert set(got) == set(ref) == {h for h, _ in records}
    problems = jplace_diff.diff_strict(got, ref)
    assert not [p for p in problems if "like_weight_ratio" in p or "rows vs" in p], problems[:5]
    assert len(problems) <= 6, problems[:5]   # rows whose scores agree to float32 rounding may swap
    for name in ("read_1", "dup_a"):
        for x, y in zip(got[name], ref[name]):
            assert x["likelihood"] == pytest.approx(y["likelihood"], rel=2e-6)
