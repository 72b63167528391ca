#!/usr/bin/env python3
"""Copies what `tools/profile_round.sh <tag>` left under gpurun_out/<tag>/ into profiles/ as r05_* and
rewrites profiles/traffic.json from the PMC summaries, stamped with the hash of the kernel sources in
the tree (run it on the same sources the GPU run used).

    python tools/collect_profiles.py r05
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epik_amd import provenance  # noqa: E402

ROUND = "r05"
FILES = {
    "bench.json": f"{ROUND}_bench.json",
    "bench_headline.json": f"{ROUND}_bench_headline_trace_run.json",
    "bench_k11.json": f"{ROUND}_bench_k11.json",
    "bench_n9999.json": f"{ROUND}_bench_n9999_team.json",
    "bench_n9999_clades.json": f"{ROUND}_bench_clades_n9999.json",
    "bench_n2999.json": f"{ROUND}_bench_n2999_team2.json",
    "bench_kmer_shard_n9999_0of8.json": f"{ROUND}_bench_kmer_shard_n9999_shard_0_of_8.json",
    "bench_amino_k7_wide_filter.json": f"{ROUND}_bench_amino_k7_1g_wide_filter.json",
    "kernel_stats_n9999_clades.csv": f"{ROUND}_kernel_stats_clades_n9999.csv",
    "kernel_stats_n2999.csv": f"{ROUND}_kernel_stats_n2999_team2.csv",
    "kernel_stats_kmer_shard_n9999_0of8.csv": f"{ROUND}_kernel_stats_kmer_shard_n9999_shard_0_of_8.csv",
    "kernel_stats_amino_k7_wide_filter.csv": f"{ROUND}_kernel_stats_amino_k7_1g_wide_filter.csv",
    "pmc_summary_n9999_clades.txt": f"{ROUND}_pmc_summary_clades_n9999.txt",
    "shard_halves_wave_timeline.txt": f"{ROUND}_shard_halves_wave_timeline.txt",
    "shard_halves_0of8_wave_timeline.txt": f"{ROUND}_shard_halves_shard_0_of_8_wave_timeline.txt",
    "team_stream_wave_timeline_clades.txt": f"{ROUND}_team_stream_wave_timeline_clades.txt",
    "bench_amino_k7.json": f"{ROUND}_bench_amino_k7_1g.json",
    "bench_kmer_shard_n9999.json": f"{ROUND}_bench_kmer_shard_n9999_1gpu.json",
    "bench_kmer_shard_n9999_256k.json": f"{ROUND}_bench_kmer_shard_n9999_1gpu_256k_reads.json",
    "kernel_stats_headline.csv": f"{ROUND}_kernel_stats.csv",
    "kernel_stats_k11.csv": f"{ROUND}_kernel_stats_k11.csv",
    "kernel_stats_n9999.csv": f"{ROUND}_kernel_stats_n9999_team.csv",
    "kernel_stats_amino_k7.csv": f"{ROUND}_kernel_stats_amino_k7_1g.csv",
    "kernel_stats_kmer_shard_n9999.csv": f"{ROUND}_kernel_stats_kmer_shard_n9999.csv",
    "kernel_stats_kmer_shard_n9999_256k.csv": f"{ROUND}_kernel_stats_kmer_shard_n9999_256k_reads.csv",
    "pmc_summary_headline.txt": f"{ROUND}_pmc_summary.txt",
    "pmc_summary_k11.txt": f"{ROUND}_pmc_summary_k11.txt",
    "pmc_summary_n9999.txt": f"{ROUND}_pmc_summary_n9999_team.txt",
    "pmc_summary_amino_k7.txt": f"{ROUND}_pmc_summary_amino_k7_1g.txt",
    "e2e_driver.txt": f"{ROUND}_e2e_driver.txt",
    "shard_rate.txt": f"{ROUND}_shard_rate_place_sharded.txt",
    "sq_counters_headline.txt": f"{ROUND}_sq_counters.txt",
    "sq_counters_n9999_team.txt": f"{ROUND}_sq_counters_n9999_team.txt",
    "sq_counters_k11.txt": f"{ROUND}_sq_counters_k11.txt",
    "sq_counters_amino_k7.txt": f"{ROUND}_sq_counters_amino_k7_1g.txt",
    "team_instruction_counts.txt": f"{ROUND}_team_instruction_counts.txt",
    "team_stream_wave_timeline.txt": f"{ROUND}_team_stream_wave_timeline.txt",
    "sweep_tree_sizes_passes.txt": f"{ROUND}_sweep_tree_sizes_passes.txt",
}


def counters(path, kernel):
    out = {}
    for line in open(path):
        parts = line.split()
        if len(parts) >= 4 and parts[0] == "bench" and parts[1] == kernel:
            out[parts[2]] = float(parts[3].split("=")[1])
    return out


def valu_busy(path, kernel):
    """Share of the kernel's cycles in which a SIMD's vector ALU is executing: SQ_ACTIVE_INST_VALU counts quad-cycles
    of VALU work summed over the 1024 SIMDs, SQ_BUSY_CYCLES the kernel's cycles summed over the 32 shader engines:
    ACTIVE x 4 / 1024 / (BUSY / 32) = ACTIVE / (8 x BUSY)."""
    try:
        c = counters(path, kernel)
        return c["SQ_ACTIVE_INST_VALU"] / (8.0 * c["SQ_BUSY_CYCLES"])
    except (OSError, KeyError, ZeroDivisionError):
        return None


def main():
    tag = sys.argv[1]
    src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
    for a, b in FILES.items():
        path = os.path.join(src, a)
        if not os.path.exists(path):
            print("missing", a)
            continue
        text = open(path).read()
        if a.endswith(".json"):
            text = text.strip().splitlines()[-1] + "\n"
        open(os.path.join(dst, b), "w").write(text)
    head = json.loads(open(os.path.join(dst, f"{ROUND}_bench.json")).read())
    old = json.load(open(os.path.join(dst, "traffic.json")))
    c = counters(os.path.join(src, "pmc_summary_headline.txt"), "place_reads_kernel")
    doc = {
        "workload": head["config"]["workload"],
        "kernel": head["roofline"]["kernel"],
        "hbm_bytes_per_launch": c["TCC_EA0_RDREQ_128B_sum"] * 128.0,
        "kernel_source_sha": provenance.kernel_source_hash(),
        "commit": subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip(),
        "method": (f"rocprofv3 --pmc in separate passes (tools/pmc_passes.sh via tools/profile_round.sh {tag}): "
                   f"TCC_EA0_RDREQ_128B_sum = {c['TCC_EA0_RDREQ_128B_sum']:.6g} requests x 128 B per launch (every request is "
                   f"a 128-B line: 32B = {c['TCC_EA0_RDREQ_32B_sum']:.0f}, 64B = {c['TCC_EA0_RDREQ_64B_sum']:.0f}); cross-check "
                   f"FETCH_SIZE = {c['FETCH_SIZE']:.6g} KiB, which on gfx950 counts 128-B requests as 64 B "
                   f"(MI355X_MICROARCH.md, HBM section) -> x2 x 1024 = {c['FETCH_SIZE'] * 2048:.4g} B; calibration streams of "
                   "tools/calib_fetch.hip in the same summary file"),
        "l2_hit_rate": c["TCC_HIT_sum"] / c["TCC_REQ_sum"],
        "valu_busy": valu_busy(os.path.join(src, "sq_counters_headline.txt"), "place_reads_kernel"),
        "source": f"profiles/{ROUND}_pmc_summary.txt",
        "note": "bench.py reports these numbers only while kernel_source_sha equals the hash of the kernel sources it "
                "runs (epik_amd/provenance.py); `workloads` is keyed by the exact config.workload string of the bench line",
        "workloads": {},
        "previous": dict(old.get("previous", {}), **{"r03 final (headline)": old.get("hbm_bytes_per_launch")}),
    }
    for bench_file, kerns, f, out, sq in (("bench_k11.json", ("place_reads_kernel",), "pmc_summary_k11.txt",
                                           f"{ROUND}_pmc_summary_k11.txt", "sq_counters_k11.txt"),
                                          ("bench_n9999.json", ("team_front_kernel", "team_stream_kernel", "team_merge_kernel"),
                                           "pmc_summary_n9999.txt", f"{ROUND}_pmc_summary_n9999_team.txt", "sq_counters_n9999_team.txt"),
                                          ("bench_n9999_clades.json", ("team_front_kernel", "team_stream_kernel", "team_merge_kernel"),
                                           "pmc_summary_n9999_clades.txt", f"{ROUND}_pmc_summary_clades_n9999.txt", "sq_counters_n9999_clades.txt"),
                                          ("bench_amino_k7.json", ("place_reads_kernel",), "pmc_summary_amino_k7.txt",
                                           f"{ROUND}_pmc_summary_amino_k7_1g.txt", "sq_counters_amino_k7.txt")):
        try:
            line = json.loads(open(os.path.join(src, bench_file)).read().strip().splitlines()[-1])
            per_kernel = [counters(os.path.join(src, f), kern) for kern in kerns]
        except (OSError, ValueError, IndexError):
            print("missing", bench_file, "or", f)
            continue
        total = lambda key: sum(c.get(key, 0.0) for c in per_kernel)  # noqa: E731
        if not total("TCC_REQ_sum"):
            print("no counters in", f)
            continue
        doc["workloads"][line["config"]["workload"]] = {
            "hbm_bytes_per_launch": total("TCC_EA0_RDREQ_128B_sum") * 128.0 + total("TCC_EA0_RDREQ_64B_sum") * 64.0
                                    + total("TCC_EA0_RDREQ_32B_sum") * 32.0,
            "l2_hit_rate": total("TCC_HIT_sum") / total("TCC_REQ_sum"), "kernels": list(kerns), "source": "profiles/" + out,
            # (of the kernel that takes the time: the one-wavefront kernel, or the streaming kernel of the three)
            "valu_busy": valu_busy(os.path.join(src, sq), kerns[0] if len(kerns) == 1 else "team_stream_kernel")}
    json.dump(doc, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print("traffic:", doc["hbm_bytes_per_launch"], "sha", doc["kernel_source_sha"], "workloads", len(doc["workloads"]))


if __name__ == "__main__":
    main()
