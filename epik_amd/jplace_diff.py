#!/usr/bin/env python3
"""Compares two jplace files by sequence name.

Strict mode (default) is the parity bar of BASELINE.json: for every name the rows must
carry the same `edge_num` in the same order -- rows whose likelihood ties exactly are
compared as sets, since the reference's std::partial_sort leaves their order unspecified
(place.cpp:153-156) -- and |delta like_weight_ratio| <= 1e-5.  Rows whose LWR sits within
the tolerance of the `keep_factor` cut may be present in one file only.

`--legacy` reproduces what the reference's scripts/jplace_diff.py measures (that script's
CLI is broken: its option is --only-best but its parameter is only_first): per name, the
two placements match if their sets of likelihood values agree within 1e-4, else if every edge
of either file is in both with |10**l1 - 10**l2| <= 1e-4; like_weight_ratio is not looked at
(jplace_diff.py:21, 197-225).  tests/test_host_cpu.py checks this mode against the reference's
own function wherever /root/reference is present.
"""
from __future__ import annotations

import sys

import click

from .jplace import read_jplace

LWR_TOL = 1e-5
LEGACY_EPS = 1e-4


def diff_strict(a: dict, b: dict, lwr_tol: float = LWR_TOL, keep_factor: float = 0.01):
    problems = []
    for name in sorted(set(a) | set(b)):
        if name not in a or name not in b:
            problems.append(f"{name}: present in one file only")
            continue
        ra, rb = a[name], b[name]
        if len(ra) != len(rb):
            # a row within tolerance of the filter cut may flip
            shorter, longer = (ra, rb) if len(ra) < len(rb) else (rb, ra)
            best = longer[0]["like_weight_ratio"] if longer else 0.0
            extra = longer[len(shorter):]
            if not all(abs(r["like_weight_ratio"] - best * keep_factor) <= lwr_tol for r in extra):
                problems.append(f"{name}: {len(ra)} rows vs {len(rb)} rows")
                continue
            ra, rb = ra[:len(shorter)], rb[:len(shorter)]
        i = 0
        while i < len(ra):
            j = i
            while j + 1 < len(ra) and ra[j + 1]["likelihood"] == ra[i]["likelihood"]:
                j += 1
            ea = sorted(r["edge_num"] for r in ra[i:j + 1])
            eb = sorted(r["edge_num"] for r in rb[i:j + 1])
            if ea != eb:
                problems.append(f"{name}: rows {i}..{j}: edges {ea} vs {eb}")
            i = j + 1
        for k, (x, y) in enumerate(zip(ra, rb)):
            if abs(x["like_weight_ratio"] - y["like_weight_ratio"]) > lwr_tol:
                problems.append(f"{name}: row {k}: like_weight_ratio {x['like_weight_ratio']} vs "
                                f"{y['like_weight_ratio']}")
    return problems


def diff_legacy(a: dict, b: dict, only_best: bool = False):
    """Semantics of the reference's scripts/jplace_diff.py (:160-172, :187-231); returns the names of
    the first file that do not match.  Per name: with `only_best`, the first rows' edges must agree
    (two empty lists match); otherwise the two SETS of likelihood values agreeing pairwise within
    1e-4 is a match whatever edges carry them (:201-208), else every edge of the union must be in both
    files with |10^l1 - 10^l2| <= 1e-4 (:212-226).  like_weight_ratio is never looked at."""
    mismatches = []
    for name in a:
        ra, rb = a[name], b[name]   # a name missing from the second file is a KeyError there too (:190)
        if only_best:
            if len(ra) == len(rb) == 0:
                continue
            if not ra or not rb or ra[0]["edge_num"] != rb[0]["edge_num"]:
                mismatches.append(name)
            continue
        la = sorted({r["likelihood"] for r in ra})
        lb = sorted({r["likelihood"] for r in rb})
        if len(la) == len(lb) and all(abs(x - y) <= LEGACY_EPS for x, y in zip(la, lb)):
            continue
        ea = {r["edge_num"]: r["likelihood"] for r in ra}
        eb = {r["edge_num"]: r["likelihood"] for r in rb}
        if any(e not in ea or e not in eb or abs(10 ** ea[e] - 10 ** eb[e]) > LEGACY_EPS for e in set(ea) | set(eb)):
            mismatches.append(name)
    return mismatches


@click.command()
@click.option("--legacy", is_flag=True, help="Semantics of the reference's scripts/jplace_diff.py.")
@click.option("--only-best", is_flag=True, help="(legacy) compare only the first placement.")
@click.option("--lwr-tol", type=float, default=LWR_TOL, show_default=True)
@click.argument("jplace1", type=click.Path(exists=True))
@click.argument("jplace2", type=click.Path(exists=True))
def main(legacy, only_best, lwr_tol, jplace1, jplace2):
    a, b = read_jplace(jplace1), read_jplace(jplace2)
    if legacy:
        bad = diff_legacy(a, b, only_best)
        print(f"{len(bad)} of {len(set(a) & set(b))} sequences differ")
        for name in bad[:50]:
            print(" ", name)
    else:
        bad = diff_strict(a, b, lwr_tol)
        print(f"{len(bad)} differences over {len(set(a) | set(b))} sequences")
        for line in bad[:50]:
            print(" ", line)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
