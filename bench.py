#!/usr/bin/env python3
"""bench.py -- reads placed / second on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (`epik::placer::place`'s per-read loop,
reference place.cpp:201-440) over one batch of synthetic reads that is already
resident in HBM.  Workload = BASELINE.json configs[1]: ~1k-branch nucl DB
(N=999, k=10, omega=1.5, mu=1.0, SURVEY.md 8d synthetic model), 150 bp reads,
1M reads per step per GPU.  With N>1 (launched by torch.distributed.run, one rank
per GPU) the reads are sharded across ranks and the DB is replicated: there is no
data-path collective (weak scaling); torch.distributed is used for the barriers
and the max-over-ranks time only.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     -- achieved algorithmic GB/s of the placement kernel (HIP events on
                  the launch stream, inside the timed region) against the HBM peak;
  cpu_baseline -- the CPU oracle (a restatement of the reference loop; the
                  reference itself cannot be built here) timed on a bounded sample
                  with all host threads.  A reported baseline, not the target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads-per-step", type=int, default=1_000_000,
                    help="reads per step per GPU (BASELINE configs[1]: 1M)")
    ap.add_argument("--read-length", type=int, default=150)
    ap.add_argument("--leaves", type=int, default=500, help="tree leaves; N = 2*leaves - 1")
    ap.add_argument("--kmer-size", type=int, default=10)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0,
                    help="target CPU time of the bounded cpu_baseline sample (0 = skip)")
    ap.add_argument("--scattered", action="store_true",
                    help="non-contiguous branch sets in the synthetic posting lists")
    ap.add_argument("--states", choices=["nucl", "amino"], default="nucl",
                    help="amino: the protein path (BASELINE configs[3]: --states amino --kmer-size 7 "
                         "--read-length 300 --p-present 0.0026); the default line is configs[1]")
    ap.add_argument("--p-present", type=float, default=0.6, help="fraction of k-mer codes that have a posting list")
    ap.add_argument("--mode", choices=["reads", "kmer-shard"], default="reads",
                    help="reads (default): reads sharded over the GPUs, database replicated, no collective.  "
                         "kmer-shard (BASELINE configs[4]: --leaves 5000 --reads-per-step 4096): rank g holds the "
                         "lists of the codes with code %% G == g, every rank accumulates ALL reads of the step, "
                         "one all-to-all + sum of the per-read branch vectors, then each rank finishes its reads")
    return ap.parse_args()


def load_traffic(workload: str):
    """HBM bytes per launch from the committed PMC pass (profiles/traffic.json), if it
    was measured for this workload; rocprofv3 --pmc cannot run inside this process."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            doc = json.load(fh)
        if doc.get("workload") == workload:
            return doc.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None


def host_cores() -> int:
    """Cores this process may really use: OMP_NUM_THREADS if set, else the smallest of the
    affinity mask and the cgroup CPU quota (the GPU boxes expose every host core through
    os.cpu_count() but schedule a one-GPU job on its 16-core share)."""
    env = int(os.environ.get("OMP_NUM_THREADS", 0) or 0)
    if env > 0:
        return env
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(db, data, offs, target_seconds: float):
    """Times the oracle (kind "port") on a bounded prefix of the same read batch."""
    from oracle import oracle
    oracle.build()
    orc = oracle.Oracle.from_synth(db)
    threads = min(host_cores(), oracle.Oracle.max_threads())
    n_total = len(offs) - 1
    probe = min(4000 * threads, n_total)
    t0 = time.perf_counter()
    orc.place(data[:int(offs[probe])], offs[:probe + 1], num_threads=threads)
    rate = probe / max(time.perf_counter() - t0, 1e-6)
    # bounded sample: the step batch, repeated until about target_seconds of wall time
    n = int(min(n_total, max(probe, rate * target_seconds)))
    reps = max(1, int(round(rate * target_seconds / n)))
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.place(data[:int(offs[n])], offs[:n + 1], num_threads=threads)
    dt = time.perf_counter() - t0
    n *= reps
    return {"value": n / dt, "unit": "reads/s", "cores": threads, "kind": "port",
            "sample": f"{n} reads (the step batch, repeated), {dt:.1f} s wall, oracle/epik_oracle.c "
                      f"(OpenMP dynamic schedule as place.cpp:218-230, CSR lookup)"}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    import torch

    from epik_amd import capi, dist as edist, synth
    from epik_amd.placer import Placer

    if not torch.cuda.is_available() or capi.device_count() == 0:
        raise SystemExit("bench.py needs a HIP device: epik_amd has no CPU fallback")
    # Rehearsal of the N>1 path on a one-GPU box (EPIK_AMD_BENCH_REHEARSAL=1): every rank uses
    # device 0 and the ranks meet over gloo, since RCCL wants one device per rank.  Its numbers
    # mean nothing; it exists so that the multi-rank code path can be run before the driver does.
    rehearsal = os.environ.get("EPIK_AMD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = edist.init_process_group("gloo" if rehearsal else "nccl")  # nccl = RCCL; None when WORLD_SIZE == 1

    # ---- synthetic workload, identical DB on every rank ---------------
    tree = synth.make_tree(args.leaves, seed=42)
    if rank == 0:
        print("[bench] building the synthetic database ...", file=sys.stderr, flush=True)
    db = synth.make_db(tree.num_nodes, states=args.states, kmer_size=args.kmer_size, seed=43,
                       p_present=args.p_present, scattered=args.scattered)
    data, offs = synth.make_reads(args.reads_per_step, args.read_length, states=args.states, seed=44 + rank)
    unit = "bp" if args.states == "nucl" else "aa"
    workload = (f"{args.states} k={args.kmer_size} omega=1.5 mu=1.0 synthetic DB, N={tree.num_nodes} branches, "
                f"{db.num_entries} postings ({db.num_entries * 8 / 1e6:.0f} MB), "
                f"{args.reads_per_step} x {args.read_length} {unit} reads per step per GPU"
                + (", scattered branch sets" if args.scattered else "")
                + (f", {args.p_present:g} of the codes present" if args.p_present != 0.6 else ""))

    if rank == 0:
        print(f"[bench] {db.num_entries} postings; uploading ...", file=sys.stderr, flush=True)
    kmer_shard = args.mode == "kmer-shard"
    if kmer_shard:  # every rank holds the same reads; --reads-per-step is the whole job's batch
        data, offs = synth.make_reads(args.reads_per_step, args.read_length, states=args.states, seed=44)
        workload += f"; k-mer-space shard over {world} GPU(s), {args.reads_per_step} reads per step in total"
    placer = Placer.from_synth(db, device=local_rank, shard_index=rank if kmer_shard else 0,
                               shard_count=world if kmer_shard else 1)
    placer.choose_counts(args.read_length)  # what epik_amd_placer_place would pick for this batch
    n = args.reads_per_step
    keep = placer.keep_at_most
    dev = torch.device("cuda", local_rank)
    d_seqs = torch.from_numpy(data).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_rows = torch.zeros(n * keep * 2, dtype=torch.float64, device=dev)   # 16 B per row
    d_nrows = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()

    def step():
        placer.place_device(d_seqs.data_ptr(), d_offs.data_ptr(), n, d_rows.data_ptr(),
                            d_nrows.data_ptr(), 0, stream.cuda_stream)

    if kmer_shard:
        N = placer.num_branches
        per = -(-n // world)
        begin, end = edist.owner_bounds(n, rank, world)
        part = [torch.zeros((per * world, N), dtype=t, device=dev) for t in (torch.float32, torch.int16)]
        recv = [torch.empty_like(x) for x in part]

        def step():  # noqa: F811
            placer.accumulate_device(d_seqs.data_ptr(), d_offs.data_ptr(), n, part[0].data_ptr(),
                                     part[1].data_ptr(), stream.cuda_stream)
            totals = part
            if dist is not None:  # one link per peer, then a sum in rank order (epik_amd.dist.place_kmer_sharded)
                totals = []
                for x, r in zip(part, recv):
                    dist.all_to_all_single(r, x)
                    totals.append(r.view(world, per, N).sum(dim=0))
            if end > begin:
                placer.finish_device(d_offs.data_ptr() + 8 * begin, end - begin, totals[0].data_ptr(),
                                     totals[1].data_ptr(), d_rows.data_ptr(), d_nrows.data_ptr(), 0,
                                     stream.cuda_stream)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()

    # ---- timed region: exactly K steps, HIP events around every launch ------------------
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        starts[i].record(stream)
        step()
        stops[i].record(stream)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = edist.max_over_ranks(elapsed, dist, device=None if rehearsal else dev)
    kernel_ms = float(np.mean([s.elapsed_time(e) for s, e in zip(starts, stops)]))

    if rank == 0:
        alg_bytes = placer.algorithmic_bytes(d_seqs.data_ptr(), d_offs.data_ptr(), n,
                                             d_nrows.data_ptr(), stream.cuda_stream)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        info = placer.launch_info()
        n_rows_host = d_nrows.cpu().numpy()
        result = {
            "metric": "reads placed/sec",
            "value": (1 if kmer_shard else world) * n * args.steps / elapsed,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if kmer_shard else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "reads_per_step_per_gpu": n,
                       "parallelism": (f"k-mer space sharded over {world} GPU(s), one all-to-all + sum per step"
                                       if kmer_shard else
                                       f"reads sharded over {world} GPU(s), DB replicated, no collective"),
                       "launch": info, "mean_rows_per_read": float(n_rows_host.mean())},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": load_traffic(workload),
                         "kernel": "place_reads_kernel", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "algorithmic_bytes_per_read": alg_bytes / n},
        }
        if args.cpu_baseline_seconds > 0 and world == 1:
            result["cpu_baseline"] = cpu_baseline(db, data, offs, args.cpu_baseline_seconds)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)

    barrier()
    placer.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
