"""Randomised differential test: many small random databases and read sets -- every k the device
supports for the alphabet, trees from a handful to thousands of branches, list lengths from 1 to the
whole tree, reads from shorter than k to several passes long, with and without ambiguous and invalid
characters -- placed on the GPU through the C ABI and by the CPU oracle.  Bit-exact bar as everywhere."""
import os

import numpy as np
import pytest

from conftest import assert_rows_match, select_kernel
from epik_amd import synth

pytestmark = pytest.mark.gpu

NUCL = "ACGT"
NUCL_AMB = "ACGTNRYKMSWBDHVU-acgtn*"
AMINO = "RHKDESTNQCGPAILMFWYV"
AMINO_AMB = AMINO + "BZJX*-"


def _random_case(seed):
    rng = np.random.default_rng(seed)
    amino = seed % 4 == 3
    if amino:
        k = int(rng.integers(1, 5))           # 20^4 = 160 000 codes at most here
        alphabet, dirty = AMINO, AMINO_AMB
    else:
        k = int(rng.choice([1, 2, 3, 5, 6, 8, 9, 10]))
        alphabet, dirty = NUCL, NUCL_AMB
    leaves = int(rng.choice([2, 3, 9, 40, 150, 700, 2100]))
    tree = synth.make_tree(leaves, seed=seed)
    db = synth.make_db(tree.num_nodes, states="amino" if amino else "nucl", kmer_size=k, seed=seed + 1,
                       p_present=float(rng.choice([0.05, 0.4, 0.9, 1.0])),
                       lognormal=(float(rng.choice([0.5, 2.0, 4.0, 6.0])), float(rng.choice([0.5, 1.5]))),
                       scattered=bool(rng.integers(0, 2)))
    reads = []
    for _ in range(int(rng.integers(20, 300))):
        kind = rng.integers(0, 10)
        if kind == 0:
            length = int(rng.integers(0, k + 2))                       # around the "shorter than k" edge
        elif kind == 1:
            length = int(rng.integers(500, 2500))                      # several passes of three tiles
        else:
            length = int(rng.integers(k, 260))
        letters = dirty if rng.integers(0, 4) == 0 else alphabet
        reads.append("".join(rng.choice(list(letters), size=length)))
    reads += ["", alphabet[0] * k, alphabet[-1] * 200]
    return db, synth.pack_reads(reads)


@pytest.mark.parametrize("layout", ["paired", "packed", "compact", "team4", "team2", "team4-sparse", "team8x2", "team4x2-classic", "team4-smallpool", "paired-fewblocks", "team4x2-fewblocks"])
@pytest.mark.parametrize("seed", range(int(os.environ.get("EPIK_AMD_RANDOM_SEEDS", "16"))))
def test_random_database_and_reads(gpu_available, oracle_lib, seed, layout, monkeypatch):
    assert gpu_available
    from epik_amd.placer import Placer
    select_kernel(monkeypatch, layout)
    db, (data, offs) = _random_case(1000 + seed)
    keep = int(np.random.default_rng(seed).choice([1, 3, 7, 12]))
    factor = float(np.random.default_rng(seed + 7).choice([0.0, 0.01, 0.5]))
    ref = oracle_lib.Oracle.from_synth(db, keep_at_most=keep, keep_factor=factor).place(data, offs, num_threads=0)
    with Placer.from_synth(db, keep_at_most=keep, keep_factor=factor) as pl:
        got = pl.place_packed(data, offs)
    assert_rows_match(*got, *ref)
