#!/usr/bin/env python3
"""Rate of epik_amd_placer_place_sharded (the --db-shard path of the driver): host reads in, rows out, G handles
holding one database cut in G by k-mer code -- here all on device 0, so the figures are what ONE device spends on G
shards' work (accumulate of all reads G times over 1/G of the lists each, the copies, the finish), not a G-device rate.

    python tools/shard_rate.py [--leaves 5000] [--reads 262144] [--shards 1 2 4]
"""
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
    ap.add_argument("--leaves", type=int, default=5000)
    ap.add_argument("--reads", type=int, default=262144)
    ap.add_argument("--shards", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--kmer-size", type=int, default=10)
    args = ap.parse_args()
    from epik_amd import synth
    from epik_amd.placer import Placer
    tree = synth.make_tree(args.leaves, seed=42)
    db = synth.make_db(tree.num_nodes, kmer_size=args.kmer_size, seed=43)
    data, offs = synth.make_reads(args.reads, 150, seed=44)
    with Placer.from_synth(db) as one:
        one.place_packed(data, offs)
        t0 = time.perf_counter()
        ref = one.place_packed(data, offs)
        t_one = time.perf_counter() - t0
    print(json.dumps({"path": "epik_amd_placer_place (one handle, the whole database)", "reads": args.reads,
                      "reads_per_s": args.reads / t_one}), flush=True)
    for G in args.shards:
        placers = [Placer.from_synth(db, shard_index=g, shard_count=G) for g in range(G)]
        try:
            Placer.place_sharded(placers, data, offs)
            times = []
            for _ in range(3):
                t0 = time.perf_counter()
                got = Placer.place_sharded(placers, data, offs)
                times.append(time.perf_counter() - t0)
            same = bool((got[0]["branch"][:, 0] == ref[0]["branch"][:, 0]).mean() > 0.999)
            print(json.dumps({"path": "epik_amd_placer_place_sharded", "shards": G, "devices": "all on device 0",
                              "reads": args.reads, "reads_per_s": args.reads / min(times), "best_rows_agree": same}), flush=True)
        finally:
            for p in placers:
                p.close()


if __name__ == "__main__":
    main()
