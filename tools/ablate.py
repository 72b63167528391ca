#!/usr/bin/env python3
"""Timing experiments on the placement kernel (developer tool, not product).

Runs the bench workload through several variants IN ONE PROCESS, interleaved over
rounds (devices differ by several percent: never compare across runs).  A variant is a
comma-separated list of key=value:
    lib=<suffix>     epik_amd/libepik_amd<suffix>.so   (e.g. lib=_ablate, lib=_exp1; default: the product lib)
    layout=compact|packed|paired, kernel=wave|team4|team8, wide=0|1|2, front=0|1 (team placement as one kernel | front + streaming + merge kernels), grid=<percent of the resident workgroups>,
    ablate=<bitmask>, stamps=1   (env read at placer creation)
LEAVES=<n> sets the tree (N = 2n - 1), N_READS the batch, CLADES=1 the workload of bench.py --clades.
Example: tools/ablate.py lib=_ablate,layout=compact lib=_exp,layout=compact
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from epik_amd import alphabet, capi, synth

    n = int(os.environ.get("N_READS", 1_000_000))
    rounds = int(os.environ.get("ROUNDS", 3))
    tree = synth.make_tree(int(os.environ.get("LEAVES", 500)), seed=42)
    if os.environ.get("CLADES"):  # bench.py --clades: lists over the clades of references, reads cut from them
        db, refs, _ = synth.make_clade_db(tree.num_nodes, kmer_size=10, seed=47)
        data, offs = synth.make_clade_reads(refs, n, 150, seed=48)
    else:
        db = synth.make_db(tree.num_nodes, kmer_size=10, seed=43)
        data, offs = synth.make_reads(n, 150, seed=44)
    dev = torch.device("cuda", 0)
    d_seqs = torch.from_numpy(data).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_rows = torch.zeros(n * 7 * 2, dtype=torch.float64, device=dev)
    d_nrows = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    off32 = np.ascontiguousarray(db.offsets, dtype=np.uint32)
    cls = alphabet.char_class_table("nucl")
    variants = [dict(x.split("=") for x in v.split(",")) for v in sys.argv[1:]] or [{}]
    placers = []
    for kv in variants:
        os.environ["EPIK_AMD_ABLATE"] = kv.get("ablate", "0")
        os.environ["EPIK_AMD_WIDE_COUNTS"] = kv.get("wide", "0")
        os.environ["EPIK_AMD_STAMPS"] = kv.get("stamps", "0")
        os.environ["EPIK_AMD_LAYOUT"] = kv.get("layout", "paired")
        os.environ.pop("EPIK_AMD_KERNEL", None)
        if "kernel" in kv:
            os.environ["EPIK_AMD_KERNEL"] = kv["kernel"]
        os.environ["EPIK_AMD_TEAM_FRONT"] = kv.get("front", "1")
        os.environ["EPIK_AMD_GRID_PERCENT"] = kv.get("grid", "0")
        os.environ.pop("EPIK_AMD_TEAM_SPARSE", None)
        if "sparse" in kv:  # 0 | always | <chunks>: the slice epilogue over the touched quads (team_epilogue.hpp)
            os.environ["EPIK_AMD_TEAM_SPARSE"] = kv["sparse"]
        os.environ["EPIK_AMD_FRONT_PER_CU"] = kv.get("fb", "0")  # workgroups per CU of the front / merge kernel (0: as queried)
        os.environ["EPIK_AMD_MERGE_PER_CU"] = kv.get("mb", "0")
        lib = ctypes.CDLL(os.path.join(ROOT, "epik_amd", f"libepik_amd{kv.get('lib', '')}.so"))
        desc = capi.PlacerDesc(
            abi_version=capi.ABI_VERSION, kmer_size=10, alphabet_size=4, num_branches=tree.num_nodes, keep_at_most=7,
            offset_bits=32, keep_factor=0.01, threshold=float(db.threshold), log_threshold=float(db.log_threshold),
            num_keys=db.num_keys, num_entries=db.num_entries, offsets=off32.ctypes.data,
            values=db.values.ctypes.data, char_class=cls.ctypes.data, device=0, shard=0)
        h = ctypes.c_void_p()
        lib.epik_amd_placer_create.argtypes = [ctypes.POINTER(capi.PlacerDesc), ctypes.POINTER(ctypes.c_void_p)]
        rc = lib.epik_amd_placer_create(ctypes.byref(desc), ctypes.byref(h))
        assert rc == 0, rc
        lib.epik_amd_placer_place_device.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_uint64] + [ctypes.c_void_p] * 4
        lib.epik_amd_placer_destroy.argtypes = [ctypes.c_void_p]
        placers.append((lib, h))

    def run(lib, h, steps):
        for _ in range(steps):
            lib.epik_amd_placer_place_device(h, d_seqs.data_ptr(), d_offs.data_ptr(), n, d_rows.data_ptr(),
                                             d_nrows.data_ptr(), None, stream)
        torch.cuda.synchronize()

    times = [[] for _ in placers]
    for lib, h in placers:
        run(lib, h, 2)
    for _ in range(rounds):
        for i, (lib, h) in enumerate(placers):
            t0 = time.perf_counter()
            run(lib, h, 4)
            times[i].append((time.perf_counter() - t0) / 4 * 1e3)
    for kv, t in zip(variants, times):
        name = ",".join(f"{k}={v}" for k, v in kv.items())
        print(f"{name:48s} min {min(t):7.3f}  median {sorted(t)[len(t) // 2]:7.3f} ms/step  "
              f"{n / min(t) / 1e3:7.2f} M reads/s", flush=True)
    for lib, h in placers:
        lib.epik_amd_placer_destroy(h)


if __name__ == "__main__":
    main()
