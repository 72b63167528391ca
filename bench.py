#!/usr/bin/env python3
"""bench.py -- reads placed / second on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (`epik::placer::place`'s per-read loop,
reference place.cpp:201-440) over one batch of synthetic reads that is already
resident in HBM.  Workload = BASELINE.json configs[1]: ~1k-branch nucl DB
(N=999, k=10, omega=1.5, mu=1.0, SURVEY.md 8d synthetic model), 150 bp reads,
1M reads per step per GPU.  With N>1 (one rank per GPU under torch.distributed.run:
either the caller launches it that way, or -- `python bench.py --gpus N` with no
WORLD_SIZE in the environment -- this script starts the launcher itself as a child
process and passes its output and exit code through) the reads are sharded across
ranks and the DB is replicated: there is no data-path collective (weak scaling);
torch.distributed is used for the barriers and the max-over-ranks time only.

Rank 0 prints ONE JSON line.  Extra objects (rank 0, N=1 where they cost time):
  roofline              -- achieved algorithmic GB/s of the placement kernel (HIP events on the
                           launch stream, inside the timed region) against the HBM peak; `bound`
                           is "hbm+mall" when the device image of the database (`working_set_bytes`)
                           is at most twice the 256 MiB Infinity Cache (the headline workload: 285 MB,
                           mostly served from it), "hbm" otherwise;
                           `traffic` = HBM bytes of the committed PMC pass, null unless that pass
                           was taken on exactly these kernel sources (profiles/traffic.json);
  roofline_hbm_resident -- the same kernel, reads and tree on a database that does NOT fit the
                           Infinity Cache (k = 11: 148 M postings, 1.2 GB), a short second pass;
  cpu_baseline          -- the CPU oracle (a restatement of the reference loop; the reference itself
                           cannot be built here) timed on a bounded sample with all host threads, run
                           the way the reference's driver runs it: batches of 2000 reads, per-batch
                           dedup, hash-map lookup, OpenMP dynamic loop; the
                           direct-index variant beside it.  A reported baseline, not the target;
  cpu_baseline_1thread  -- the same on one thread;
  e2e                   -- FASTA in -> jplace closed through the native driver epik-dna, the
                           reference's own "Placement time" quantity (main.cpp:322,378-381), on the step's
                           million reads, timed in microseconds;
  host_entry            -- the boundary's own rate: reads/s through `epik_amd_placer_place` (pageable host buffers in,
                           rows out, synchronous: what a drop-in placer::place calls, place.h:103) on the step's
                           reads, and under `large_tree` on the N = 9 999 tree; the box's raw pageable copy rates beside it;
  roofline_large_tree, roofline_large_tree_clades
                        -- the tree of BASELINE configs[4] (N = 9 999) placed in one pass by the front / streaming /
                           merge kernels, three steps: SURVEY 8d's random lists, and lists over the clades of
                           reference sequences with reads cut from them (synth.make_clade_db);
  kmer_shard_1gpu, kmer_shard_0of8_1gpu, kmer_shard_rank_of8_1gpu
                        -- the two halves of the k-mer-space-sharded placement of that tree on this one GPU
                           (65 536 reads per step, a process of its own): the whole database as one shard;
                           shard 0 of 8 alone with ALL reads finished from that one source; and the same shard as one
                           rank of eight works -- all reads accumulated, an eighth finished from eight sources;
  build                 -- compiler, kernel-source hash and ISA-lint record of the library (epik_amd/provenance.py).
config.world_size is what the process group reports; config.ranks lists every rank's device ordinal, PCI bus id,
host and own ms_per_step (the k-mer-space shard: the bytes it sent per step) -- a SCALE record says by itself
whether N ranks on N devices took part.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import re
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MALL_BYTES = 256 << 20  # Infinity Cache


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads-per-step", type=int, default=None,
                    help="reads per step per GPU (default: BASELINE configs[1]'s 1M; --mode kmer-shard: 65536 -- the whole "
                         "job's batch, which every rank accumulates in full, its partial lists inside 32-bit offsets)")
    ap.add_argument("--read-length", type=int, default=150)
    ap.add_argument("--leaves", type=int, default=500, help="tree leaves; N = 2*leaves - 1")
    ap.add_argument("--kmer-size", type=int, default=10)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0,
                    help="target CPU time of the bounded cpu_baseline sample (0 = skip the CPU legs)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the HBM-resident second pass and the end-to-end driver run")
    ap.add_argument("--scattered", action="store_true",
                    help="non-contiguous branch sets in the synthetic posting lists")
    ap.add_argument("--clades", action="store_true",
                    help="a database shaped like one built from reference sequences (synth.make_clade_db: the k-mers of a "
                         "reference carry lists over the reference's clade) and reads cut from the references: a read adds "
                         "k-mer after k-mer into the same rows")
    ap.add_argument("--states", choices=["nucl", "amino"], default="nucl",
                    help="amino: the protein path (BASELINE configs[3]: --states amino --kmer-size 7 "
                         "--read-length 300 --p-present 0.0026); the default line is configs[1]")
    ap.add_argument("--p-present", type=float, default=0.6, help="fraction of k-mer codes that have a posting list")
    ap.add_argument("--shard-of", type=int, default=0, metavar="G",
                    help="--mode kmer-shard on ONE GPU: the database holds shard 0 of G only (the lists of the codes with "
                         "code %% G == 0) -- what one of G GPUs accumulates per batch; the rows that come out are those of "
                         "that shard alone (a timing experiment: DESIGN.md 6)")
    ap.add_argument("--as-rank", action="store_true",
                    help="with --shard-of G: what ONE of G GPUs computes per batch -- accumulate of ALL reads against its shard, "
                         "then the finish of its n / G reads from G sources (all of them this shard's lists: the rows mean "
                         "nothing, the work is a rank's; the exchange is not in it)")
    ap.add_argument("--mode", choices=["reads", "kmer-shard"], default="reads",
                    help="reads (default): reads sharded over the GPUs, database replicated, no collective.  "
                         "kmer-shard (BASELINE configs[4]: --leaves 5000 --reads-per-step 4096): rank g builds and "
                         "holds only the lists of the codes with code %% G == g, every rank accumulates ALL reads of "
                         "the step, one all-to-all + sum of the per-read branch vectors, then each rank finishes its reads")
    args = ap.parse_args()
    if args.reads_per_step is None:
        args.reads_per_step = 65536 if args.mode == "kmer-shard" else 1_000_000
    return args


def load_counters(workload: str) -> dict:
    """What the committed counter passes say about this workload (profiles/traffic.json: one entry per workload
    string: HBM bytes per launch, share of cycles the vector ALUs are busy) -- only if they were measured ON THESE
    KERNEL SOURCES (rocprofv3 --pmc cannot run inside this process)."""
    from epik_amd import provenance
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            doc = json.load(fh)
        if doc.get("kernel_source_sha") != provenance.kernel_source_hash():
            return {}
        return doc if doc.get("workload") == workload else (doc.get("workloads", {}).get(workload) or {})
    except (OSError, ValueError):
        pass
    return {}


# What random 128-byte line reads get from a working set far larger than the Infinity Cache (tools/probe_mall.hip:
# 6.3 TB/s on 4 GB with four consecutive lines per wave; tools/probe_lines.hip: 6.3 TB/s on 16 GB with a line per
# LANE, the shape of a lookup): the bound of a workload made of table lookups rather than of posting streams.
RANDOM_LINE_GBPS = 6300.0


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
    """Times the oracle (kind "port") on a bounded prefix of the same read batch, the way the
    reference's driver runs placer::place: batches of 2000 reads, per-batch dedup,
    OpenMP dynamic loop; lookup through a node-chained hash map (the reference's data structure
    shape) and, beside it, through the direct index.  Median of five runs each."""
    from oracle import oracle
    oracle.build()
    orc = oracle.Oracle.from_synth(db)
    threads = min(host_cores(), oracle.Oracle.max_threads())
    n_total = len(offs) - 1

    def timed(n, n_threads, runs=5):
        times = []
        for _ in range(runs):
            t0 = time.perf_counter()
            orc.place_batched(data[:int(offs[n])], offs[:n + 1], batch_size=2000, num_threads=n_threads)
            times.append(time.perf_counter() - t0)
        return float(np.median(times))

    out = {}
    for label, n_threads, share in (("all", threads, 0.65), ("one", 1, 0.35)):
        per_variant = target_seconds * share / 2 / 5  # two variants, five runs each
        variants = {}
        for name, use_hash in (("hash_map", True), ("direct_index", False)):
            orc.use_hash_map(use_hash)
            probe = min(2000 * n_threads, n_total)
            rate = probe / max(timed(probe, n_threads, runs=1), 1e-6)
            n = int(min(n_total, max(probe, rate * per_variant)))
            variants[name] = {"reads_per_s": n / timed(n, n_threads), "reads": n}
        out[label] = {
            "value": variants["hash_map"]["reads_per_s"], "unit": "reads/s", "cores": n_threads, "kind": "port",
            # (the box's cores beside the share this job may use: `value` x box_cores / cores is what the whole
            # host would do if the loop scaled perfectly)
            "box_cores": os.cpu_count(),
            "variants": {k: v["reads_per_s"] for k, v in variants.items()},
            "sample": f"{variants['hash_map']['reads']} reads of the step batch, median of 5 runs, "
                      "oracle/epik_oracle.c run as the reference's driver runs placer::place: batches of 2000 reads, "
                      "per-batch dedup, OpenMP dynamic loop (place.cpp:201-275); `value` = the hash-map lookup variant "
                      "(node-chained map key -> vector of postings), `variants.direct_index` = CSR lookup"}
    orc.use_hash_map(False)
    return out["all"], out["one"]


def end_to_end(db, tree, data, read_length: int, n_reads: int, jobs: int):
    """FASTA in -> jplace closed through epik_amd/bin/epik-dna (the reference's "Placement time",
    main.cpp:322,378-381: after the database is loaded, FASTA reading and jplace writing included)."""
    from epik_amd import dbfile
    binary = os.path.join(ROOT, "epik_amd", "bin", "epik-dna" if db.states == "nucl" else "epik-aa")
    if not os.path.exists(binary):
        return {"reads_per_s": None, "note": f"{binary} not built (make -C epik_amd/host)"}
    tmp = tempfile.mkdtemp(prefix="epik_bench_e2e_")
    try:
        db_path, fasta = os.path.join(tmp, "db.ekdb"), os.path.join(tmp, "reads.fasta")
        dbfile.write_db(db_path, db, tree.newick())
        seqs = data[:n_reads * read_length].reshape(n_reads, read_length)
        with open(fasta, "wb") as fh:
            for s0 in range(0, n_reads, 50_000):
                block = seqs[s0:s0 + 50_000]
                lines = np.empty((len(block), read_length + 1), dtype=np.uint8)
                lines[:, :read_length] = block
                lines[:, read_length] = ord("\n")
                heads = [b">r%d\n" % i for i in range(s0, s0 + len(block))]
                fh.write(b"".join(h + row.tobytes() for h, row in zip(heads, lines)))
        out_dir = os.path.join(tmp, "out")
        os.makedirs(out_dir)
        run = subprocess.run([binary, "-d", db_path, "-q", fasta, "-o", out_dir, "-j", str(jobs)],
                             capture_output=True, text=True, timeout=600, env=dict(os.environ, EPIK_AMD_STAGE_TIMES="1"))
        m = re.search(r"Placement time: .*\((\d+) ms\)", run.stdout)
        if run.returncode != 0 or not m:
            return {"reads_per_s": None, "note": (run.stdout + run.stderr)[-400:]}
        ms = max(int(m.group(1)), 1)
        us = re.search(r"placement_time_us (\d+)", run.stdout)   # (the reference's line is in whole milliseconds)
        us = int(us.group(1)) if us else ms * 1000
        jplace = os.path.join(out_dir, "placements_reads.fasta.jplace")
        result = {"reads_per_s": n_reads / (us / 1e6), "reads": n_reads, "placement_time_ms": ms, "placement_time_us": us, "jobs": jobs,
                  "batch_size": 2000, "jplace_mb": os.path.getsize(jplace) / 1e6,
                  "what": "epik-dna: FASTA parse -> per-batch dedup -> GPU placement -> jplace written and closed "
                          "(database load excluded, as the reference's timer)"}
        stages = dict(re.findall(r"stage (read|place|write) ([0-9.]+) ms", run.stdout))
        if stages:
            result["stage_busy_ms"] = {k: float(v) for k, v in stages.items()}
            if float(stages.get("write", 0)) > 0:
                result["writer_gbps"] = os.path.getsize(jplace) / (float(stages["write"]) * 1e-3) / 1e9
        # (one writer formats and writes the jplace of every device's batches: FASTA -> jplace does not scale with
        # devices, the device-resident rate does -- DESIGN.md 4)
        result["scales_with_devices"] = False
        result["cpu_baseline"] = end_to_end_cpu(db, tree, tmp, fasta, n_reads, jobs)
        return result
    except (OSError, subprocess.SubprocessError) as e:
        return {"reads_per_s": None, "note": repr(e)[:400]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def end_to_end_cpu(db, tree, tmp, fasta, n_reads: int, threads: int):
    """The CPU twin of `end_to_end` (timing (b)): oracle/epik_oracle_driver -- the restatement run
    as the reference's driver runs, FASTA batch -> placer::place (dedup, hash-map lookup, OpenMP loop) -> jplace
    appended, read and write on the main thread -- on the same database and the same reads, all host threads.
    Kind "port": the reference's own driver cannot be built here."""
    from epik_amd.placer import pendant_lengths
    driver = os.path.join(ROOT, "oracle", "epik_oracle_driver")
    try:
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "epik_oracle_driver"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        with open(os.path.join(tmp, "tree.txt"), "w") as fh:
            fh.write(tree.newick(jplace=True))
        distal, pendant = pendant_lengths(tree.branch_length, tree.subtree_num_nodes, tree.subtree_total_length)
        np.stack([distal, pendant], 1).astype(np.float64).tofile(os.path.join(tmp, "lengths.f64"))
        out = os.path.join(tmp, "cpu.jplace")
        run = subprocess.run([driver, os.path.join(tmp, "db.ekdb"), fasta, out, os.path.join(tmp, "tree.txt"),
                              os.path.join(tmp, "lengths.f64"), str(threads)], capture_output=True, text=True, timeout=900)
        m = re.search(r"Placement time: (\d+) ms", run.stdout)
        if run.returncode != 0 or not m:
            return {"value": None, "note": (run.stdout + run.stderr)[-300:]}
        ms = max(int(m.group(1)), 1)
        return {"value": n_reads / (ms / 1e3), "unit": "reads/s", "cores": threads, "kind": "port",
                "placement_time_ms": ms, "jplace_mb": os.path.getsize(out) / 1e6,
                "sample": f"the same {n_reads} reads and database through oracle/epik_oracle_driver: FASTA -> jplace "
                          "closed, batches of 2000, hash-map lookup, OpenMP loop, reading and writing on the main "
                          "thread as main.cpp:332-361"}
    except (OSError, subprocess.SubprocessError) as e:
        return {"value": None, "note": repr(e)[:300]}


def host_entry(pl, data, offs, n: int, reps: int = 3):
    """The boundary's own rate: `epik_amd_placer_place` -- what a drop-in for placer::place (place.h:103: host records
    in, placed_collection out) calls -- with PAGEABLE host buffers in and out, synchronous: copy in, kernels, copy out,
    pipelined in chunks by the library.  Best of `reps` calls after one warm-up; the raw pageable copy rates of this
    box beside it (one torch copy each way of the same buffers)."""
    import ctypes
    import torch
    from epik_amd import capi
    keep = pl.keep_at_most
    rows = np.ones((n, keep), dtype=capi.PLACEMENT)   # ones: the pages exist before the timing starts
    n_rows = np.ones(n, dtype=np.uint32)
    counts = np.ones((n, keep), dtype=np.uint32)
    seqs = np.ascontiguousarray(data[:int(offs[n])])
    seq_offs = np.ascontiguousarray(offs[:n + 1], dtype=np.uint64)

    def call():
        capi.check(pl._lib.epik_amd_placer_place(pl._handle, seqs.ctypes.data, seq_offs.ctypes.data, n, rows.ctypes.data,
                                                 n_rows.ctypes.data, counts.ctypes.data))
    call()
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        call()
        times.append(time.perf_counter() - t0)
    best = min(times)
    bytes_in, bytes_out = seqs.nbytes + seq_offs.nbytes, rows.nbytes + n_rows.nbytes + counts.nbytes
    dev = torch.device("cuda", pl.device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d = torch.from_numpy(seqs).to(dev)
    torch.cuda.synchronize()
    h2d = seqs.nbytes / (time.perf_counter() - t0) / 1e9
    d_rows = torch.empty(rows.nbytes, dtype=torch.uint8, device=dev)
    back = torch.from_numpy(rows.view(np.uint8).reshape(-1))   # (pages that exist: the copy alone is timed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    back.copy_(d_rows)
    torch.cuda.synchronize()
    d2h = rows.nbytes / (time.perf_counter() - t0) / 1e9
    del d, d_rows, back
    return {"entry_point": "epik_amd_placer_place (pageable host buffers in, rows out, synchronous)", "reads": n,
            "reads_per_s": n / best, "ms": best * 1e3, "ms_all": [t * 1e3 for t in times],
            "bytes_in": bytes_in, "bytes_out": bytes_out, "moved_gbps": (bytes_in + bytes_out) / best / 1e9,
            "pageable_h2d_gbps": h2d, "pageable_d2h_gbps": d2h,
            "rows_checksum": int(n_rows.astype(np.int64).sum())}


def pci_bus_id(device: int) -> str | None:
    """PCI bus id of a HIP device ("0000:c1:00.0"), from torch's device properties where they carry it, else from the
    HIP runtime this process holds."""
    import ctypes
    import torch
    try:
        props = torch.cuda.get_device_properties(device)
        if hasattr(props, "pci_bus_id"):
            return "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, getattr(props, "pci_device_id", 0))
    except Exception:  # noqa: BLE001 -- evidence only
        pass
    try:
        from epik_amd import capi
        runtimes = capi.hip_runtimes()
        if runtimes:
            hip = ctypes.CDLL(runtimes[0])
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) == 0:
                return buf.value.decode()
    except Exception:  # noqa: BLE001
        pass
    return None


def sub_bench(extra_args, timeout=600):
    """Another bench.py line in a process of its own (the k-mer-space shard on this GPU): the parsed JSON, or a note."""
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_ENV}
    try:
        run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-seconds", "0", "--no-extras"]
                             + list(extra_args), capture_output=True, text=True, timeout=timeout, env=env)
        lines = [line for line in run.stdout.splitlines() if line.startswith("{")]
        if run.returncode != 0 or not lines:
            return {"value": None, "note": (run.stdout + run.stderr)[-400:]}
        return json.loads(lines[-1])
    except (OSError, ValueError, subprocess.SubprocessError) as err:
        return {"value": None, "note": repr(err)[:400]}


def image_bytes(plan) -> int:
    return int(plan.table_bytes + plan.filter_bytes + plan.posting_bytes)


def kernel_name(plan, pl, placing: bool = True) -> str:
    """What the HIP events around a launch time, as the handle says it ran (`epik_amd_placer_last_path`, not
    what the plan would have liked): one kernel, or -- on a large tree -- the kernels of the team placement
    back to back on the stream (epik_amd/csrc/team_stream.hip)."""
    from epik_amd import capi
    path = pl.last_path()
    if path == capi.PATH_WAVE:
        return "place_reads_kernel" if placing else "place_reads_kernel (accumulate) + finish_reads_kernel"
    w = plan.team_waves
    if path == capi.PATH_TEAM_STREAMED:
        if placing:
            return f"team_front_kernel<{w}> + team_stream_kernel<{w}> + team_merge_kernel"
        return (f"accumulate: team_front_kernel<{w}> + team_sparse_scan_kernel + team_stream_kernel<{w}>; "
                f"finish: team_header_kernel + team_stream_kernel<{w}> + team_merge_kernel")
    return f"team_place_kernel<{w}>"


LAUNCHER_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE",
                "TORCHELASTIC_RUN_ID")


def self_launch(n_ranks: int) -> int:
    """`python bench.py --gpus N` with no launcher around it: start `python -m torch.distributed.run --nproc-per-node N
    bench.py <the same arguments>` as a CHILD process -- this process has not touched HIP or torch yet and never will
    (a process that has initialised the GPU must not exec; a child is always allowed) --, pass the child's output
    through (rank 0's one JSON line on stdout, the progress lines on stderr) and return its exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_ENV}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on these hosts
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n_ranks)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n_ranks} without a launcher: starting {' '.join(cmd[1:8])} ... as a child process", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, cwd=ROOT)
    try:
        return child.wait()
    except KeyboardInterrupt:
        child.terminate()
        return child.wait()


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    import torch

    from epik_amd import alphabet, capi, dist as edist, placer as eplacer, provenance, synth
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
    kmer_shard = args.mode == "kmer-shard"

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    # ---- synthetic workload --------------------------------------------------
    tree = synth.make_tree(args.leaves, seed=42)
    log("building the synthetic database ...")
    # k-mer-space shard: rank g builds and keeps the lists of the codes with code % G == g and never holds
    # the others -- they are empty lists in its descriptor (include/epik_amd.h, "k-mer-space shard")
    # (a key space too large to draw one number per code -- amino k = 7: 1.28 G codes -- gets its present codes drawn
    # directly, and goes to create() in the sparse form of the descriptor: keys[present] + offsets, ABI 3)
    big_key_space = alphabet.alphabet_size(args.states) ** args.kmer_size > (1 << 28)
    clade_refs = None
    if args.clades:
        if kmer_shard or args.states != "nucl":
            raise SystemExit("--clades: nucleotide databases, reads-sharded mode")
        db, clade_refs, _ = synth.make_clade_db(tree.num_nodes, kmer_size=args.kmer_size, seed=47)
    elif big_key_space and not kmer_shard:
        db = synth.make_sparse_db(tree.num_nodes, states=args.states, kmer_size=args.kmer_size, seed=43,
                                  p_present=args.p_present, dense=False)
        db.total_entries = db.num_entries
    else:
        db = synth.make_db(tree.num_nodes, states=args.states, kmer_size=args.kmer_size, seed=43,
                           p_present=args.p_present, scattered=args.scattered,
                           shard=((0, args.shard_of) if args.shard_of else (rank, world)) if kmer_shard else None)
    total_entries = db.total_entries
    if clade_refs is not None:
        data, offs = synth.make_clade_reads(clade_refs, args.reads_per_step, args.read_length, seed=48 + rank)
    else:
        data, offs = synth.make_reads(args.reads_per_step, args.read_length, states=args.states,
                                      seed=44 if kmer_shard else 44 + rank)
    unit = "bp" if args.states == "nucl" else "aa"
    workload = (f"{args.states} k={args.kmer_size} omega=1.5 mu=1.0 synthetic DB, N={tree.num_nodes} branches, "
                f"{total_entries} postings ({total_entries * 8 / 1e6:.0f} MB), "
                f"{args.reads_per_step} x {args.read_length} {unit} reads per step per GPU"
                + (", scattered branch sets" if args.scattered else "")
                + (f", shard 0 of {args.shard_of} only" if args.shard_of and kmer_shard else "")
                + (f", as one rank of {args.shard_of}: all reads accumulated, {args.reads_per_step // args.shard_of} finished from "
                   f"{args.shard_of} sources" if args.as_rank and kmer_shard else "")
                + (f", {args.p_present:g} of the codes present" if args.p_present != 0.6 and not args.clades else "")
                + (", lists over the clades of 500 references of 1500 bp, reads cut from the references (1 % substitutions)"
                   if args.clades else ""))
    if kmer_shard:  # every rank holds the same reads; --reads-per-step is the whole job's batch
        workload += f"; k-mer-space shard over {world} GPU(s), {args.reads_per_step} reads per step in total"

    log(f"{db.num_entries} postings on this rank; uploading ...")
    plan = eplacer.plan(db)
    placer = Placer.from_synth(db, device=local_rank)
    placer.choose_counts(args.read_length)  # what epik_amd_placer_place would pick for this batch
    n = args.reads_per_step
    keep = placer.keep_at_most
    dev = torch.device("cuda", local_rank)
    d_seqs = torch.from_numpy(data).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_rows = torch.zeros(n * keep * 2, dtype=torch.float64, device=dev)   # 16 B per row
    d_nrows = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()

    def make_step(pl):
        def step():
            pl.place_device(d_seqs.data_ptr(), d_offs.data_ptr(), n, d_rows.data_ptr(),
                            d_nrows.data_ptr(), 0, stream.cuda_stream)
        return step

    step = make_step(placer)

    drain = None
    shard_info = None
    if kmer_shard:
        N = placer.num_branches
        if args.as_rank and not (args.shard_of and world == 1):
            raise SystemExit("--as-rank: with --shard-of G on one GPU")
        n_parts = args.shard_of if args.as_rank else world   # finishers the batch is divided among
        per = -(-n // n_parts)
        begin, end = edist.owner_bounds(n, 0 if args.as_rank else rank, n_parts)
        pinfo = placer.partial_info()
        shard_info = {"partials": "lists" if pinfo["lists"] else "dense"}
    if kmer_shard and pinfo["lists"]:
        # ---- partial lists (large trees): per read and slice only the rows this shard's lists touched ------
        S, eb = pinfo["slices"], pinfo["entry_bytes"]
        comm = torch.cuda.Stream(dev) if (dist is not None and not rehearsal) else None

        def alloc(cap):
            return {"entries": torch.empty(max(cap, 1) * eb, dtype=torch.uint8, device=dev), "cap": cap,
                    "index": torch.zeros((per * n_parts, S, 2), dtype=torch.int32, device=dev),
                    "part_entries": torch.zeros(n_parts, dtype=torch.int64, device=dev), "done": None}

        def accumulate(buf):
            if buf.get("finished") is not None:   # (the finish that read this set of buffers last)
                stream.wait_event(buf["finished"])
            placer.accumulate_lists_device(d_seqs.data_ptr(), d_offs.data_ptr(), n, n_parts, buf["entries"].data_ptr(),
                                           buf["cap"], buf["index"].data_ptr(), buf["part_entries"].data_ptr(),
                                           stream.cuda_stream)
            buf["done"] = stream.record_event()

        # one untimed pass says how much room the parts take (the same batch every step)
        want = int(n * max(args.read_length - args.kmer_size + 1, 1) * pinfo["postings_per_kmer"] * 2) + 65536
        if want >= (1 << 32):
            raise SystemExit(f"--reads-per-step {n}: the partial lists of one step would take {want} entries, and offsets inside "
                             f"a part are 32-bit (include/epik_amd.h); at most {int(n * ((1 << 32) - 65536) / want)} reads per step "
                             "with this database (the default of this mode is 65536)")
        probe = alloc(want)
        accumulate(probe)
        torch.cuda.synchronize()
        need = int(probe["part_entries"].sum().item())
        if need > probe["cap"]:
            raise SystemExit(f"the sizing pass of the partial lists overflowed ({need} > {probe['cap']} entries)")
        del probe
        bufs = [alloc(need + 1024) for _ in range(2)]
        # The finish of a batch runs on a stream of its own, beside the accumulate of the next one: the two halves
        # lean on different resources (the accumulate waits on LDS round trips of the stream, the finish issues the
        # epilogue's sweeps), a CU that holds workgroups of both is busier than one that holds either (the handle
        # keeps separate headers for the two; include/epik_amd.h)
        fin_stream = torch.cuda.Stream(dev)
        shard_info.update({"entry_bytes": eb, "slices": S, "entries_per_read": need / n,
                           "partial_bytes_per_read": (need * eb + S * 8 * n) / n,
                           "dense_bytes_per_read": 6 * N})
        own = {"entries": [None] * n_parts, "index": [None] * n_parts}

        def exchange(buf):
            """all-to-all of the parts (split sizes from a gather of the part sizes) and of their index, on the
            communication stream behind THIS batch's accumulate only (epik_amd.dist._exchange_lists)."""
            ctx = torch.cuda.stream(comm) if comm is not None else contextlib.nullcontext()
            with ctx:
                if comm is not None:
                    comm.wait_event(buf["done"])
                else:
                    buf["done"].synchronize()
                staged = rehearsal
                mine = buf["part_entries"].cpu() if staged else buf["part_entries"]
                sizes = torch.empty(world * world, dtype=torch.int64, device=mine.device)
                dist.all_gather_into_tensor(sizes, mine)
                sizes = sizes.cpu().view(world, world)
                total = int(sizes[rank].sum())
                send_split = [int(x) * eb for x in sizes[rank]]
                recv_split = [int(sizes[g][rank]) * eb for g in range(world)]
                send, send_index = buf["entries"][:total * eb], buf["index"]
                # what leaves this rank per step: the parts of the other ranks' reads and their index
                shard_info["sent_bytes_per_step"] = (sum(send_split) - send_split[rank]
                                                     + send_index.numel() * send_index.element_size() * (world - 1) // world)
                if staged:
                    send, send_index = send.cpu(), send_index.cpu()
                recv = torch.empty(sum(recv_split), dtype=torch.uint8, device=send.device)
                dist.all_to_all_single(recv, send, output_split_sizes=recv_split, input_split_sizes=send_split)
                recv_index = torch.empty_like(send_index)
                dist.all_to_all_single(recv_index.view(torch.uint8).view(-1), send_index.view(torch.uint8).view(-1))
                if staged:
                    recv, recv_index = recv.to(dev), recv_index.to(dev)
                at = 0
                for g in range(world):
                    own["entries"][g] = recv[at:at + recv_split[g]]
                    own["index"][g] = recv_index.view(world, per, S, 2)[g]
                    at += recv_split[g]
                own["keep"] = (recv, recv_index)
                return comm.record_event() if comm is not None else None

        def finish_lists(buf, after):
            fin_stream.wait_event(after)
            if end > begin:
                placer.finish_lists_device(d_offs.data_ptr() + 8 * begin, end - begin,
                                           [e.data_ptr() if e.numel() else 0 for e in own["entries"]],
                                           [x.data_ptr() for x in own["index"]], d_rows.data_ptr(), d_nrows.data_ptr(),
                                           0, fin_stream.cuda_stream)
            buf["finished"] = fin_stream.record_event()

        state = {"i": 0, "pending": None}

        def complete(buf):
            if dist is None:
                # (--as-rank: part 0 -- this rank's reads -- of every one of the G shards: here G times this shard's)
                for g in range(n_parts):
                    own["entries"][g], own["index"][g] = buf["entries"], buf["index"]
                finish_lists(buf, buf["done"])
            else:
                arrived = exchange(buf)
                finish_lists(buf, arrived if arrived is not None else buf["done"])

        def step():  # noqa: F811
            # batch i accumulates while batch i - 1 crosses (several GPUs) and finishes (two sets of buffers)
            buf = bufs[state["i"] & 1]
            state["i"] += 1
            accumulate(buf)
            if state["pending"] is not None:
                complete(state["pending"])
            state["pending"] = buf

        def drain():  # noqa: F811
            if state["pending"] is not None:
                complete(state["pending"])
                state["pending"] = None
            stream.wait_event(fin_stream.record_event())   # (the timed region ends on the launch stream)
    elif kmer_shard:
        part = [torch.zeros((per * world, N), dtype=t, device=dev) for t in (torch.float32, torch.int16)]
        shard_info.update({"partial_bytes_per_read": 6 * N, "dense_bytes_per_read": 6 * N})

        def exchange(x):  # one link per peer; the rows cross as bytes (epik_amd.dist.place_kmer_sharded)
            sent = x.view(torch.uint8)
            received = torch.empty_like(sent)
            dist.all_to_all_single(received, sent)
            return received.view(x.dtype).view(world, per, N)

        def step():  # noqa: F811
            placer.accumulate_device(d_seqs.data_ptr(), d_offs.data_ptr(), n, part[0].data_ptr(),
                                     part[1].data_ptr(), stream.cuda_stream)
            totals = part
            if dist is not None:
                totals = []
                for x in part:
                    received = exchange(x)
                    total = received[0].clone()
                    for g in range(1, world):  # rank order: the float32 sums do not depend on the transport
                        total += received[g]
                    totals.append(total)
            if end > begin:
                placer.finish_device(d_offs.data_ptr() + 8 * begin, end - begin, totals[0].data_ptr(),
                                     totals[1].data_ptr(), d_rows.data_ptr(), d_nrows.data_ptr(), 0,
                                     stream.cuda_stream)

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed_steps(step_fn, steps, warmup, drain_fn=None):
        for _ in range(warmup):
            step_fn()
        if drain_fn is not None:
            drain_fn()
        torch.cuda.synchronize()
        barrier()
        starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        stops = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            starts[i].record(stream)
            step_fn()
            stops[i].record(stream)
        if drain_fn is not None:  # (a pipelined step leaves its last batch to be finished: inside the timed region)
            drain_fn()
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        return elapsed, float(np.mean([s.elapsed_time(e) for s, e in zip(starts, stops)]))

    # ---- timed region: exactly K steps, HIP events around every launch ------------------
    elapsed, kernel_ms = timed_steps(step, args.steps, args.warmup, drain)
    if drain is not None:
        # (a pipelined step: the halves of consecutive batches run side by side on two streams -- the events around a
        # step on the launch stream see one of them; the step's share of the timed region is the launch time)
        kernel_ms = elapsed / args.steps * 1e3
    # ---- what a reader of the line needs to see that N ranks on N devices really took part -------------------
    evidence = {"rank": rank, "local_rank": local_rank, "device": int(torch.cuda.current_device()),
                "pci_bus_id": pci_bus_id(int(torch.cuda.current_device())), "host": socket.gethostname(),
                "ms_per_step": elapsed / args.steps * 1e3}
    if shard_info and "sent_bytes_per_step" in shard_info:
        evidence["sent_bytes_per_step"] = int(shard_info["sent_bytes_per_step"])
    ranks = [evidence]
    if dist is not None:
        ranks = [None] * world
        dist.all_gather_object(ranks, evidence)
    elapsed = edist.max_over_ranks(elapsed, dist, device=None if rehearsal else dev)

    def roofline_of(pl, pl_plan, ms, workload_name, seqs=None, offs=None):
        seqs, offs = (d_seqs, d_offs) if seqs is None else (seqs, offs)
        alg_bytes = pl.algorithmic_bytes(seqs.data_ptr(), offs.data_ptr(), n, d_nrows.data_ptr(), stream.cuda_stream)
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        working_set = image_bytes(pl_plan)
        measured = load_counters(workload_name)
        traffic = measured.get("hbm_bytes_per_launch")
        # a working set of up to twice the Infinity Cache is served from it for a good part (the headline
        # database: 285 MB against 268 MB of cache); only well beyond that is the kernel bound by HBM alone
        roof = {"bound": "hbm+mall" if working_set <= 2 * MALL_BYTES else "hbm", "achieved": achieved,
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "working_set_bytes": working_set, "mall_bytes": MALL_BYTES,
                "kernel": kernel_name(pl_plan, pl, not kmer_shard), "kernel_ms": ms, "algorithmic_bytes_per_launch": alg_bytes,
                "algorithmic_bytes_per_read": alg_bytes / n}
        if measured.get("valu_busy") is not None:
            # what else bounds the kernel: the share of its cycles in which a SIMD's vector ALU is executing
            # (SQ_ACTIVE_INST_VALU x 4 / SIMDs / kernel cycles, the committed SQ counter pass of this workload)
            roof["valu_busy"] = measured["valu_busy"]
        if traffic and working_set > 2 * MALL_BYTES:  # (a working set the Infinity Cache mostly holds is not priced against HBM's rate)
            # The SURVEY 8(d) formula charges 8 bytes per lookup; a lookup FETCHES a 128-byte line (of the table or
            # of the presence filter).  A workload that is mostly lookups -- a sparse protein database: 294 of them
            # per read against a few found lists -- is bound by how many random lines the memory system serves, so
            # beside the algorithmic fraction: the lines really fetched (PMC) per second against that rate.
            fetched = traffic / (ms * 1e-3) / 1e9
            roof["fetched"] = {"achieved": fetched, "peak": RANDOM_LINE_GBPS, "unit": "GB/s", "frac": fetched / RANDOM_LINE_GBPS,
                               "bytes_per_read": traffic / n, "lines_per_read": traffic / n / 128.0,
                               "what": "bytes fetched past L2 (TCC_EA0_RDREQ x request size, profiles/traffic.json) per "
                                       "second against the rate of random 128-byte reads on a working set beyond the "
                                       "Infinity Cache (tools/probe_mall.hip)"}
        return roof

    if rank == 0:
        info = placer.launch_info()
        n_rows_host = d_nrows.cpu().numpy().view(np.uint32).astype(np.int64)
        if (n_rows_host > keep).any():
            raise SystemExit("a read came back marked EPIK_AMD_ROWS_COUNTS_TOO_NARROW: the bench chose too narrow counts")
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
                       "parallelism": (f"k-mer space sharded over {world} GPU(s), one exchange step per batch "
                                       f"({shard_info['partials']} partials), overlapped with the next batch's accumulate"
                                       if kmer_shard else
                                       f"reads sharded over {world} GPU(s), DB replicated, no collective"),
                       "collectives": (dist.get_backend() if dist is not None else None),
                       # (as the process group reports it, not as the command line asked)
                       "world_size": (dist.get_world_size() if dist is not None else 1), "ranks": ranks,
                       "launch": info, "mean_rows_per_read": float(n_rows_host.mean()),
                       **({"kmer_shard": shard_info} if shard_info else {})},
            "roofline": roofline_of(placer, plan, kernel_ms, workload),
            # the compiler the library was built with, the kernel sources it was built from, and whether the ISA
            # lint of the streaming loop passed on both (epik_amd/provenance.py)
            "build": provenance.summary(),
        }
    extras = rank == 0 and world == 1 and not kmer_shard
    if extras and not args.no_extras:
        # ---- the boundary's own rate (PCIe-inclusive), beside the device-resident `value` and the FASTA -> jplace `e2e`
        log("host entry point (epik_amd_placer_place) ...")
        result["host_entry"] = {"workload": workload, **host_entry(placer, data, offs, n)}
    placer.close()

    if extras and not args.no_extras and args.states == "nucl" and args.kmer_size < 11:
        # ---- the same kernel on a database the Infinity Cache cannot hold (SURVEY.md 8d sized the
        # workload to be HBM-bound; the packed layout moved the headline database just under 256 MiB)
        log("second pass: k = 11 database (1.2 GB on the device) ...")
        big = synth.make_db(tree.num_nodes, states="nucl", kmer_size=11, seed=43, p_present=args.p_present,
                            scattered=args.scattered)
        big_plan = eplacer.plan(big)
        with Placer.from_synth(big, device=local_rank) as big_placer:
            big_placer.choose_counts(args.read_length)
            _, big_ms = timed_steps(make_step(big_placer), 6, 2)
            # (the workload string of `bench.py --kmer-size 11`, which is what the PMC passes of that database ran)
            big_total = big.total_entries
            roof = roofline_of(big_placer, big_plan, big_ms,
                               f"{args.states} k=11 omega=1.5 mu=1.0 synthetic DB, N={tree.num_nodes} branches, "
                               f"{big_total} postings ({big_total * 8 / 1e6:.0f} MB), "
                               f"{args.reads_per_step} x {args.read_length} {unit} reads per step per GPU"
                               + (", scattered branch sets" if args.scattered else "")
                               + (f", {args.p_present:g} of the codes present" if args.p_present != 0.6 else ""))
        roof["workload"] = (f"nucl k=11, N={tree.num_nodes}, {big.num_entries} postings, the same "
                            f"{n} x {args.read_length} bp reads; 6 steps after 2 warm-ups")
        roof["reads_per_s"] = n / (big_ms * 1e-3)
        result["roofline_hbm_resident"] = roof
        del big
    default_workload = (args.leaves == 500 and args.kmer_size == 10 and not args.clades and not args.scattered
                        and args.p_present == 0.6 and args.read_length == 150)
    if extras and not args.no_extras and args.states == "nucl" and default_workload:
        # ---- the weakest product paths, in the driver's record: the large tree of configs[4] placed
        # in one pass on this GPU -- SURVEY.md 8d's random lists, and lists over the clades of references -- and the
        # two halves of its k-mer-space-sharded placement.  Three steps each.
        host_reads = (data, offs)   # (the step's reads on the host: what the host entry point is handed)

        def large_tree(clades):
            big_tree = synth.make_tree(5000, seed=42)
            seqs, offs = d_seqs, d_offs
            if clades:
                big_db, refs, _ = synth.make_clade_db(big_tree.num_nodes, kmer_size=args.kmer_size, seed=47)
                c_data, c_offs = synth.make_clade_reads(refs, n, args.read_length, seed=48)
                seqs, offs = torch.from_numpy(c_data).to(dev), torch.from_numpy(c_offs.view(np.int64)).to(dev)
            else:
                big_db = synth.make_db(big_tree.num_nodes, states="nucl", kmer_size=args.kmer_size, seed=43)
            big_plan = eplacer.plan(big_db)
            total = big_db.total_entries
            name = (f"nucl k={args.kmer_size} omega=1.5 mu=1.0 synthetic DB, N={big_tree.num_nodes} branches, {total} postings "
                    f"({total * 8 / 1e6:.0f} MB), {n} x {args.read_length} bp reads per step per GPU"
                    + (", lists over the clades of 500 references of 1500 bp, reads cut from the references (1 % substitutions)"
                       if clades else ""))
            with Placer.from_synth(big_db, device=local_rank) as pl:
                pl.choose_counts(args.read_length)

                def one():
                    pl.place_device(seqs.data_ptr(), offs.data_ptr(), n, d_rows.data_ptr(), d_nrows.data_ptr(), 0, stream.cuda_stream)
                _, ms = timed_steps(one, 3, 1)
                roof = roofline_of(pl, big_plan, ms, name, seqs, offs)
                if not clades:
                    result.setdefault("host_entry", {})["large_tree"] = {"workload": name, **host_entry(pl, host_reads[0], host_reads[1], n)}
            keep_keys = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_read", "valu_busy")
            out = {k: roof[k] for k in keep_keys if k in roof}
            out["reads_per_s"] = n / (ms * 1e-3)
            out["traffic_over_algorithmic"] = (roof["traffic"] / roof["algorithmic_bytes_per_launch"]) if roof.get("traffic") else None
            out["workload"] = name + "; 3 steps after 1 warm-up"
            return out
        log("large tree (N = 9999), one pass ...")
        result["roofline_large_tree"] = large_tree(False)
        log("large tree (N = 9999), lists over clades ...")
        result["roofline_large_tree_clades"] = large_tree(True)

        def shard_line(extra):
            doc = sub_bench(["--mode", "kmer-shard", "--leaves", "5000", "--steps", "5", "--warmup", "2"] + extra)
            if doc.get("value") is None:
                return doc
            shard = doc["config"].get("kmer_shard", {})
            return {"reads_per_s": doc["value"], "ms_per_step": doc["ms_per_step"], "reads_per_step": doc["config"]["reads_per_step_per_gpu"],
                    "ms_per_million_reads": doc["ms_per_step"] / doc["config"]["reads_per_step_per_gpu"] * 1e6,
                    "frac": doc["roofline"]["frac"], "achieved": doc["roofline"]["achieved"], "unit": doc["roofline"]["unit"],
                    "kernel": doc["roofline"]["kernel"], "entries_per_read": shard.get("entries_per_read"),
                    "partial_bytes_per_read": shard.get("partial_bytes_per_read"), "workload": doc["config"]["workload"]}
        log("k-mer-space shard (N = 9999) on this GPU: the whole database, then shard 0 of 8 ...")
        result["kmer_shard_1gpu"] = shard_line([])
        result["kmer_shard_0of8_1gpu"] = shard_line(["--shard-of", "8"])
        # (what one rank of eight computes per batch: the accumulate of all reads, the finish of an eighth from eight sources)
        result["kmer_shard_rank_of8_1gpu"] = shard_line(["--shard-of", "8", "--as-rank"])
    if extras and args.cpu_baseline_seconds > 0:
        log("CPU baseline (oracle) ...")
        result["cpu_baseline"], result["cpu_baseline_1thread"] = cpu_baseline(db, data, offs, args.cpu_baseline_seconds)
    elif rank == 0:
        result["cpu_baseline"] = None
    if extras and not args.no_extras:
        log("end to end through the native driver ...")
        result["e2e"] = end_to_end(db, tree, data, args.read_length, min(n, 1_000_000), host_cores())
        result["e2e_reads_per_s"] = result["e2e"].get("reads_per_s")
    if rank == 0:
        print(json.dumps(result), flush=True)

    barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
