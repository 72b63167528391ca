"""The N>1 path on CPU: world_size-2 (and 3) gloo runs of the read-sharding glue
(epik_amd/dist.py) that bench.py and multi-GPU callers use."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from epik_amd import dist as edist, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 1000, 1001):
        for world in (1, 2, 3, 8):
            spans = [edist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


def test_shard_reads_rebases_offsets():
    data, offs = synth.pack_reads(["ACGT", "AC", "", "GGGTT", "T"])
    parts = [edist.shard_reads(data, offs, r, 2) for r in range(2)]
    assert bytes(parts[0][0]) == b"ACGTAC" and list(parts[0][1]) == [0, 4, 6]
    assert bytes(parts[1][0]) == b"GGGTTT" and list(parts[1][1]) == [0, 0, 5, 6]
    assert parts[0][2] == (0, 2) and parts[1][2] == (2, 5)


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sharded_placement_matches_single_process(world, oracle_lib):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert f"dist ok: world={world}" in out.stdout


def test_owner_bounds_are_equal_slices():
    for n in (0, 1, 7, 203, 1000):
        for world in (1, 2, 3, 8):
            spans = [edist.owner_bounds(n, r, world) for r in range(world)]
            per = -(-n // world) if n else 0
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(b - a <= per for a, b in spans) and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_kmer_sharded_placement_matches_oracle(world, oracle_lib):
    """The exchange step of the k-mer-space shard: all-to-all of the per-read branch vectors +
    sum in rank order, numpy engine in place of the two kernel halves; then the same batch in three pieces
    with partial lists (epik_amd.dist.place_kmer_sharded_lists: variable-size all-to-all, index, overflow)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker_kmer.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert f"kmer-shard ok: world={world}" in out.stdout
    # ... and with partial lists: three batches through the pipelined exchange, an overflow round among them
    for b in range(3):
        assert f"kmer-shard lists batch {b} ok: world={world}" in out.stdout, out.stdout[-2000:]


def test_gloo_kmer_sharded_lists_when_one_rank_overflows_alone(oracle_lib):
    """A rank whose lists found no room must not leave the exchange by itself (the others would go on to the
    all-to-all while it waits in another collective): the part sizes cross together with every rank's capacity,
    and all ranks repeat the accumulate together."""
    world = 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker_kmer.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1", EPIK_AMD_TEST_ONE_RANK_OVERFLOWS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    for b in range(3):
        assert f"kmer-shard lists batch {b} ok: world={world}" in out.stdout, out.stdout[-2000:]
