#!/usr/bin/env python3
"""Summarises the counter_collection.csv files written by tools/pmc_passes.sh:
per kernel of interest, the mean of every counter over its dispatches."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv"))):
    run = f.split(os.sep)[-3]
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        for key in ("place_reads_kernel", "team_place_kernel", "team_front_kernel", "team_stream_kernel",
                    "team_merge", "stream_seq", "stream_rnd"):
            if key in name:
                # (team_merge_kernel, or since round 5 team_merge_packed_kernel<16 | 32>: one line either way)
                acc[(run.split("_")[0], "team_merge_kernel" if key == "team_merge" else key, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (run, kern, ctr), vals in sorted(acc.items()):
    print(f"{run:6s} {kern:20s} {ctr:28s} mean={sum(vals) / len(vals):.6g}  n={len(vals)}")
