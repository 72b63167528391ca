#!/usr/bin/env python3
"""host_rate.py -- PCIe-inclusive rate of the host-buffer entry point
(`epik_amd_placer_place`: copy in, kernel, copy out, synchronous) on the bench workload,
next to the device-resident rate `bench.py` reports.  DESIGN.md section 4 quotes it.

    python tools/host_rate.py [--reads N] [--reps R]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--leaves", type=int, default=500)
    ap.add_argument("--kmer-size", type=int, default=10)
    args = ap.parse_args()

    from epik_amd import capi, synth
    from epik_amd.placer import Placer

    tree = synth.make_tree(args.leaves, seed=42)
    db = synth.make_db(tree.num_nodes, kmer_size=args.kmer_size, seed=43)
    data, offs = synth.make_reads(args.reads, 150, seed=44)
    placer = Placer.from_synth(db)
    n, keep = args.reads, placer.keep_at_most
    rows = np.ones((n, keep), dtype=capi.PLACEMENT)       # ones: pages touched before timing
    n_rows = np.ones(n, dtype=np.uint32)
    counts = np.ones((n, keep), dtype=np.uint32)
    lib, handle = placer._lib, placer._handle

    def call():
        capi.check(lib.epik_amd_placer_place(handle, data.ctypes.data, offs.ctypes.data, n,
                                             rows.ctypes.data, n_rows.ctypes.data, counts.ctypes.data))

    call()
    times = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        call()
        times.append(time.perf_counter() - t0)
    bytes_in = data.nbytes + offs.nbytes
    bytes_out = rows.nbytes + n_rows.nbytes + counts.nbytes
    best = min(times)
    print(json.dumps({"entry_point": "epik_amd_placer_place (host buffers)", "reads": n,
                      "ms_best": best * 1e3, "ms_all": [t * 1e3 for t in times],
                      "reads_per_s": n / best, "bytes_in": bytes_in, "bytes_out": bytes_out,
                      "checksum_rows": int(n_rows.sum())}))
    placer.close()


if __name__ == "__main__":
    main()
