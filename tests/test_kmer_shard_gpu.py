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


@pytest.fixture(autouse=True, params=["paired", "compact", "team4", "team8", "team4-classic", "team4-smallpool", "paired-fewblocks", "team4-fewblocks"])
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
