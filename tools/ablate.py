#!/usr/bin/env python3
"""Timing experiments on the placement kernel (developer tool, not product):
runs the bench workload through variants of the kernel selected by environment
variables read at placer creation (EPIK_AMD_LDS_ATOMIC, and -- with the
`make -C epik_amd/csrc ablate` library -- EPIK_AMD_ABLATE)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from epik_amd import synth
    from epik_amd.placer import Placer

    n = int(os.environ.get("N_READS", 1_000_000))
    tree = synth.make_tree(500, seed=42)
    db = synth.make_db(tree.num_nodes, kmer_size=10, seed=43)
    data, offs = synth.make_reads(n, 150, seed=44)
    dev = torch.device("cuda", 0)
    d_seqs = torch.from_numpy(data).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_rows = torch.zeros(n * 7 * 2, dtype=torch.float64, device=dev)
    d_nrows = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    variants = [v.split(",") for v in sys.argv[1:]] or [["atomic=1", "ablate=0"]]
    for var in variants:
        kv = dict(x.split("=") for x in var)
        os.environ["EPIK_AMD_LDS_ATOMIC"] = kv.get("atomic", "1")
        os.environ["EPIK_AMD_ABLATE"] = kv.get("ablate", "0")
        pl = Placer.from_synth(db)
        for _ in range(2):
            pl.place_device(d_seqs.data_ptr(), d_offs.data_ptr(), n, d_rows.data_ptr(),
                            d_nrows.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 5
        for _ in range(steps):
            pl.place_device(d_seqs.data_ptr(), d_offs.data_ptr(), n, d_rows.data_ptr(),
                            d_nrows.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print(f"{','.join(var):40s} {ms:8.3f} ms/step  {n / ms / 1e3:8.2f} M reads/s  {pl.launch_info()}", flush=True)
        pl.close()


if __name__ == "__main__":
    main()
