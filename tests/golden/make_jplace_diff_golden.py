#!/usr/bin/env python3
"""Generates tests/golden/jplace_diff_reference.json: what the reference's own comparison function
(/root/reference/scripts/jplace_diff.py -- the parity tool BASELINE.json's north_star names; its click
wrapper is broken, the function behind it is callable) says about seeded pairs of jplace documents with
every kind of disagreement.  Run in the build container, where the reference is present; the fixture
holds inputs and the reference's verdicts only (data, no reference text).

    python tests/golden/make_jplace_diff_golden.py
"""
import contextlib
import importlib.util
import io
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from epik_amd import jplace  # noqa: E402

REFERENCE_DIFFER = "/root/reference/scripts/jplace_diff.py"


def make_pair(seed):
    rng = np.random.default_rng(seed)
    names = [f"s{i}" for i in range(24)]

    def rows(n):
        edges = rng.choice(40, size=n, replace=False)
        return [dict(edge_num=int(e), likelihood=float(-rng.uniform(0.5, 6.0)), like_weight_ratio=float(rng.random()),
                     distal_length=0.1, pendant_length=0.2) for e in edges]

    first = {name: rows(int(rng.integers(0, 6))) for name in names}
    second = json.loads(json.dumps(first))
    for i, name in enumerate(names):
        r = second[name]
        kind = (i + seed) % 7
        if kind == 1 and r:
            r[0]["likelihood"] += 5e-5            # inside the 1e-4 window
        elif kind == 2 and r:
            r[0]["likelihood"] -= 0.5             # outside it
        elif kind == 3 and r:
            r[0]["edge_num"] = 99                 # another edge, the same likelihood
        elif kind == 4 and r:
            r.pop()                               # a row missing
        elif kind == 5:
            r.append(dict(edge_num=77, likelihood=-9.0, like_weight_ratio=0.0, distal_length=0.0, pendant_length=0.0))
        elif kind == 6 and len(r) > 1:
            r[0], r[1] = r[1], r[0]               # the two best rows swapped
    return first, second


def write(path, placed):
    doc = {"fields": jplace.FIELDS, "version": 3, "tree": "(A:1{0},B:2{1}):0{2};", "metadata": {},
           "placements": [{"p": [[r[f] for f in jplace.FIELDS] for r in rws], "nm": [[name, 1]]}
                          for name, rws in placed.items()]}
    with open(path, "w") as fh:
        json.dump(doc, fh)


def main():
    spec = importlib.util.spec_from_file_location("reference_jplace_diff", REFERENCE_DIFFER)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(4):
            first, second = make_pair(seed)
            a, b = os.path.join(tmp, "a.jplace"), os.path.join(tmp, "b.jplace")
            write(a, first)
            write(b, second)
            for only_best in (False, True):
                out = io.StringIO()
                with contextlib.redirect_stdout(out):
                    ref.jplace_diff.callback(a, b, only_best)
                matched = int(out.getvalue().strip().splitlines()[-1].split("/")[0])
                cases.append({"seed": seed, "only_best": only_best, "first": first, "second": second,
                              "names": len(first), "reference_matched": matched})
    with open(os.path.join(HERE, "jplace_diff_reference.json"), "w") as fh:
        json.dump({"generator": "tests/golden/make_jplace_diff_golden.py",
                   "reference": "scripts/jplace_diff.py: jplace_diff(), EPSILON = 1e-4 on 10**likelihood",
                   "cases": cases}, fh)
    print(len(cases), "cases;", [c["reference_matched"] for c in cases])


if __name__ == "__main__":
    main()
