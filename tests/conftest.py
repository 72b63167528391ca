import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """Builds oracle/libepik_oracle.so with the committed Makefile (gcc only)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def small_case():
    """N=15 tree, k=4 nucl DB: small enough for the pure-Python restatement."""
    from epik_amd import synth
    tree = synth.make_tree(8, seed=1)
    db = synth.make_db(tree.num_nodes, kmer_size=4, p_present=0.7, seed=5, lognormal=(1.0, 1.0))
    return tree, db


def select_kernel(monkeypatch, name):
    """Environment of a placer for the GPU tests.  `name`: a layout of the one-wavefront kernel
    (paired, filtered, packed, compact), or teamW[xP] -- the team placement as front kernel +
    streaming kernel (team_stream.hip) --, teamW[xP]-classic -- team_place_kernel alone --, or
    teamW[xP]-smallpool -- a descriptor pool so small that some reads of a batch fall to
    team_place_kernel behind the streaming kernel; teamW[xP]-sparse / -dense -- the WIDE build of the streaming
    kernel (EPIK_AMD_STREAM_WIDE=1: by itself only large slices get it) with its slice epilogue over the touched quads
    (team_epilogue.hpp) wherever their list holds them / nowhere; teamW[xP]-block2 / -block2wide -- the streaming kernel
    in workgroups of two waves (EPIK_AMD_STREAM_BLOCK), lean / wide build.  A layout of the one-wavefront kernel with -runs: lists that
    are one ascending run of branches stored without their cells (the kernels with the run path).  Any of them with -fewblocks: a device that holds two
    workgroups (EPIK_AMD_MAX_BLOCKS), so that the waves of a test-sized batch place several reads one after
    the other on the grids of a million-read batch (capi.hip: spread_grid)."""
    for var in ("EPIK_AMD_TEAM_FRONT", "EPIK_AMD_TEAM_POOL", "EPIK_AMD_LAYOUT", "EPIK_AMD_MAX_BLOCKS", "EPIK_AMD_RUNS",
                "EPIK_AMD_TEAM_SPARSE", "EPIK_AMD_STREAM_WIDE", "EPIK_AMD_STREAM_BLOCK"):
        monkeypatch.delenv(var, raising=False)
    if name.endswith("-runs"):  # the packed lists run-coded (by itself the builder does that for large databases only)
        name = name[:-len("-runs")]
        monkeypatch.setenv("EPIK_AMD_RUNS", "1")
    if name.endswith("-fewblocks"):
        name = name[:-len("-fewblocks")]
        monkeypatch.setenv("EPIK_AMD_MAX_BLOCKS", "2")
    if name.startswith("team"):
        kernel, _, variant = name.partition("-")
        monkeypatch.setenv("EPIK_AMD_KERNEL", kernel)
        if variant == "classic":
            monkeypatch.setenv("EPIK_AMD_TEAM_FRONT", "0")
        elif variant == "smallpool":
            monkeypatch.setenv("EPIK_AMD_TEAM_POOL", "2048")
        elif variant == "sparse":  # the slice epilogue over the touched quads whatever the slice's size
            monkeypatch.setenv("EPIK_AMD_TEAM_SPARSE", "always")
            # (that epilogue lives in the WIDE build of the streaming kernel, which by itself only slices of ~2 200
            # rows and more get: forced onto the small trees of the tests -- with 8 slices per pass there is none)
            monkeypatch.setenv("EPIK_AMD_STREAM_WIDE", "1")
        elif variant == "dense":  # ... and never: the wide build with its dense epilogue
            monkeypatch.setenv("EPIK_AMD_TEAM_SPARSE", "0")
            monkeypatch.setenv("EPIK_AMD_STREAM_WIDE", "1")
        elif variant in ("block2", "block2wide"):
            # the streaming kernel in workgroups of TWO waves (by itself where LDS then holds more waves on a CU:
            # N = 2 999, 3 999, 5 999 -- never on the small trees of the tests), lean build / wide build
            monkeypatch.setenv("EPIK_AMD_STREAM_BLOCK", "2")
            if variant == "block2wide":
                monkeypatch.setenv("EPIK_AMD_STREAM_WIDE", "1")
                monkeypatch.setenv("EPIK_AMD_TEAM_SPARSE", "always")
        else:
            assert variant == "", name
    else:
        monkeypatch.setenv("EPIK_AMD_KERNEL", "wave")
        monkeypatch.setenv("EPIK_AMD_LAYOUT", name)


def mixed_reads(rng, n, k, alphabet_plain="ACGT", alphabet_amb="ACGTNRY-", max_len=60):
    reads = []
    for i in range(n):
        length = int(rng.integers(k, max_len))
        alpha = alphabet_plain if i % 3 else alphabet_amb
        reads.append("".join(rng.choice(list(alpha), size=length)))
    return reads


@pytest.fixture(scope="session")
def gpu_available():
    # torch first: it brings its own HIP runtime, which must initialise before libepik_amd's
    # (the system one) does, or torch finds "No HIP GPUs" later in this process (bench.py's order)
    import torch
    torch.cuda.is_available()
    from epik_amd import capi
    return capi.device_count() > 0


def assert_rows_match(rows_gpu, n_gpu, cnt_gpu, rows_ref, n_ref, cnt_ref, lwr_tol=1e-5):
    """The parity bar of BASELINE.json: branch order and float32 scores bit-exact,
    |delta like_weight_ratio| <= 1e-5 (north_star).  Returns max |delta lwr|."""
    assert np.array_equal(n_gpu, n_ref), (
        f"row counts differ at reads {np.nonzero(n_gpu != n_ref)[0][:10]}")
    keep = rows_ref.shape[1]
    valid = np.arange(keep)[None, :] < n_ref[:, None]
    assert np.array_equal(rows_gpu["branch"][valid], rows_ref["branch"][valid]), "branch ids differ"
    a = rows_gpu["score"][valid].view(np.uint32)
    b = rows_ref["score"][valid].view(np.uint32)
    assert np.array_equal(a, b), f"float32 scores not bit-identical ({int((a != b).sum())} rows)"
    assert np.array_equal(cnt_gpu[valid], cnt_ref[valid]), "k-mer counts differ"
    d = np.abs(rows_gpu["lwr"][valid] - rows_ref["lwr"][valid])
    worst = float(d.max()) if d.size else 0.0
    assert worst <= lwr_tol, f"|delta lwr| = {worst} > {lwr_tol}"
    return worst
