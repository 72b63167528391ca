#!/usr/bin/env python3
"""e2e_bench.py -- end-to-end rate of the native driver: FASTA in -> jplace closed
(SURVEY.md 8d asks for it beside the kernel-only rate; the reference's "Placement time",
main.cpp:322,378-381, is this quantity).

    python tools/e2e_bench.py [--reads N] [--batch-size B ...]

Writes the bench database (EPIKAMD1 container) and an N x 150 bp FASTA to a scratch
directory, runs epik_amd/bin/epik-dna on them and prints one JSON line per batch size.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_fasta(path, data, n, length):
    """`>r<i>` + one sequence line per read, assembled with numpy (no Python loop per base)."""
    seqs = data[:n * length].reshape(n, length)
    with open(path, "wb") as fh:
        step = 100_000
        for s0 in range(0, n, step):
            block = seqs[s0:s0 + step]
            out = bytearray()
            for i, row in enumerate(block):
                out += b">r%d\n" % (s0 + i)
                out += row.tobytes()
                out += b"\n"
            fh.write(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--batch-size", type=int, nargs="+", default=[2000, 100_000])
    ap.add_argument("--leaves", type=int, default=500)
    ap.add_argument("--kmer-size", type=int, default=10)
    ap.add_argument("--jobs", type=int, nargs="+", default=[1])
    ap.add_argument("--devices", default="", help="passed to the driver (e.g. 0,0: two handles on the one device)")
    ap.add_argument("--keep", action="store_true", help="keep the scratch directory")
    ap.add_argument("--binary", nargs="+", default=["epik-dna"], help="driver binaries under epik_amd/bin to run")
    args = ap.parse_args()

    from epik_amd import dbfile, synth

    subprocess.run(["make", "-C", os.path.join(ROOT, "epik_amd", "host")], check=True, stdout=subprocess.DEVNULL)
    tmp = tempfile.mkdtemp(prefix="epik_e2e_")
    tree = synth.make_tree(args.leaves, seed=42)
    db = synth.make_db(tree.num_nodes, kmer_size=args.kmer_size, seed=43)
    db_path = os.path.join(tmp, "db.ekdb")
    dbfile.write_db(db_path, db, tree.newick())
    data, _ = synth.make_reads(args.reads, 150, seed=44)
    fasta = os.path.join(tmp, "reads.fasta")
    write_fasta(fasta, data, args.reads, 150)
    for bs, binary, jobs in [(b, x, j) for b in args.batch_size for x in args.binary for j in args.jobs]:
        out_dir = os.path.join(tmp, f"out_{bs}")
        os.makedirs(out_dir, exist_ok=True)
        cmd = [os.path.join(ROOT, "epik_amd", "bin", binary), "-d", db_path, "-q", fasta, "-o", out_dir,
               "--batch-size", str(bs), "-j", str(jobs)] + (["--devices", args.devices] if args.devices else [])
        t0 = time.perf_counter()
        run = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, EPIK_AMD_STAGE_TIMES="1"))
        wall = time.perf_counter() - t0
        if run.returncode != 0:
            print(run.stdout[-1500:], run.stderr[-1500:])
            raise SystemExit(run.returncode)
        m = re.search(r"Placement time: .*\((\d+) ms\)", run.stdout)
        place_ms = int(m.group(1)) if m else None
        jp = os.path.join(out_dir, "placements_reads.fasta.jplace")
        print(json.dumps({"binary": binary, "reads": args.reads, "batch_size": bs, "jobs": jobs, "devices": args.devices or "0",
                          "placement_time_ms": place_ms,
                          "reads_per_s_fasta_to_jplace": args.reads / (place_ms / 1e3) if place_ms else None,
                          "process_wall_s": wall, "fasta_mb": os.path.getsize(fasta) / 1e6,
                          "jplace_mb": os.path.getsize(jp) / 1e6,
                          "stages": re.findall(r"^stage .*$", run.stdout, flags=re.M),
                          # (EPIK_AMD_WRITE_TIMES=1: the writer's two halves, summed over the groups)
                          "writer_ms": {"format": round(sum(float(x) for x in re.findall(r"format ([\d.]+) ms", run.stderr)), 1),
                                        "to_file": round(sum(float(x) for x in re.findall(r"(?:pwrite|mapped copy) ([\d.]+) ms", run.stderr)), 1),
                                        "how": os.environ.get("EPIK_AMD_JPLACE_WRITE", "pwrite")}
                          if os.environ.get("EPIK_AMD_WRITE_TIMES") else None}), flush=True)
    if not args.keep:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
