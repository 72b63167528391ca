#!/usr/bin/env python3
"""Writes tests/golden/synth_k6.json.

The reference holds no golden vectors for this path, cannot be
built in this image, and is not importable (it is C++), so these vectors come
from the C oracle (oracle/epik_oracle.c) after cross-checking every row against
the independent numpy restatement (oracle/epik_oracle_np.py).  They freeze the
agreed outputs; they do not pin the oracle to the reference ("parity unpinned").

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from epik_amd import alphabet, synth  # noqa: E402
from oracle import oracle  # noqa: E402
from oracle.epik_oracle_np import RefShapedPlacer, dict_db_from_csr  # noqa: E402

CFG = dict(n_leaves=20, tree_seed=7, kmer_size=6, p_present=0.5, db_seed=11, lognormal=[1.5, 1.2])


def main():
    oracle.build()
    tree = synth.make_tree(CFG["n_leaves"], seed=CFG["tree_seed"])
    db = synth.make_db(tree.num_nodes, kmer_size=CFG["kmer_size"], p_present=CFG["p_present"],
                       seed=CFG["db_seed"], lognormal=tuple(CFG["lognormal"]))
    rng = np.random.default_rng(123)
    reads = []
    for i in range(120):
        length = int(rng.integers(6, 200))
        alpha = "ACGT" if i % 4 else "ACGTNRYKM-"
        reads.append("".join(rng.choice(list(alpha), size=length)))
    reads += ["ACGTA", "ACGTAC", "N" * 30, "ACGTAC" * 40, "T" * 6, "acgtnacgtacguu"]
    data, offs = synth.pack_reads(reads)
    orc = oracle.Oracle.from_synth(db)
    rows, n_rows, counts = orc.place(data, offs)
    ref = RefShapedPlacer(dict_db_from_csr(db.offsets, db.values), kmer_size=db.kmer_size,
                          alphabet_size=4, num_branches=tree.num_nodes, threshold=db.threshold,
                          log_threshold=db.log_threshold,
                          char_class=alphabet.char_class_table("nucl"))
    out_rows = []
    for i, r in enumerate(reads):
        exp = ref.place(r.encode()) or []
        assert len(exp) == n_rows[i], (i, r)
        cur = []
        for j, (b, s, lwr, c) in enumerate(exp):
            x = rows[i, j]
            assert int(x["branch"]) == b and x["lwr"] == lwr and counts[i, j] == c
            assert x["score"].view(np.uint32) == np.float32(s).view(np.uint32)
            cur.append([int(b), int(x["score"].view(np.uint32)), float(lwr), int(c)])
        out_rows.append(cur)
    doc = dict(CFG, reads=reads, n_rows=[int(x) for x in n_rows], rows=out_rows,
               fields=["branch", "score_f32_bits", "lwr", "kmer_count"],
               num_branches=tree.num_nodes, threshold_bits=int(db.threshold.view(np.uint32)))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_k6.json")
    with open(path, "w") as fh:
        json.dump(doc, fh, indent=None, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
