#!/usr/bin/env python3
"""How much of a slice a (read, slice) item touches on the synthetic database, and what a list of its touched quads
would save (developer tool, CPU only; what team_epilogue.hpp's thresholds were reasoned from):

    python tools/touched_density.py [leaves=5000] [reads=400]

Per item: chunks streamed, touched quads (4 rows).  Then, for a list of up to `cap` trips of 64 quads and a
chunk-count predictor, the instructions saved per item under the round-4 cost model (a dense trip ~77 instructions
over both sweeps, a sparse one ~110, the count pass ~15 per dense trip)."""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from epik_amd import synth  # noqa: E402


def main():
    leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    tree = synth.make_tree(leaves, seed=42)
    n, w, k = tree.num_nodes, 4, 10
    db = synth.make_db(n, kmer_size=k, seed=43)
    data, offs = synth.make_reads(n_reads, 150, seed=44)
    slice_rows = (n + w - 1) // w
    rows_pad = (slice_rows + 1 + 15) & ~15
    quads = rows_pad // 4
    dense_trips = (quads + 63) // 64
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    items = []
    for r in range(n_reads):
        v = np.array([code[c] for c in data[offs[r]:offs[r + 1]]])
        touched = np.zeros(n, bool)
        chunks = np.zeros(w, int)
        for p in range(len(v) - k + 1):
            key = 0
            for c in v[p:p + k]:
                key = key * 4 + c
            br = db.values["branch"][db.offsets[key]:db.offsets[key + 1]]
            touched[br] = True
            for sl in range(w):
                chunks[sl] += (((br // slice_rows) == sl).sum() + 63) // 64
        for sl in range(w):
            cell = rows_pad - 1 - np.nonzero(touched[sl * slice_rows:(sl + 1) * slice_rows])[0]
            items.append((chunks[sl], len(np.unique(cell // 4))))
    items = np.array(items)
    ch, q = items[:, 0], items[:, 1]
    trips = (q + 63) // 64
    print(f"N {n}: slices of {slice_rows} rows, {quads} quads, {dense_trips} dense trips; per item: {ch.mean():.1f} chunks, "
          f"{q.mean():.0f} touched quads ({q.mean() / quads:.2f})")
    for cap in (3, 4, 5):
        ok = trips <= cap
        print(f"list of up to {cap} trips: {ok.mean():.2f} of the items, {trips[ok].mean():.2f} trips each")
        for thr in (24, 32, 40, 48, 10 ** 6):
            tried = ch <= thr
            gain = np.where(tried & ok, 77 * dense_trips - 15 * dense_trips - 110 * trips, 0) - np.where(tried & ~ok, 15 * dense_trips * 0.7, 0)
            print(f"   asked when <= {thr:>7d} chunks: {tried.mean():.2f} asked, {(tried & ok).mean():.2f} sparse, "
                  f"{gain.mean():6.1f} instructions saved per item")


if __name__ == "__main__":
    main()
