#!/usr/bin/env python3
"""Summary of the single-wave timeline a diagnostic build writes (EPIK_AMD_STAMPS=1, EPIK_AMD_TRACE_FILE=<path>;
team_stream.hip / place_device.hpp): cycles per read of one wave of team_stream_kernel by section.  Every entry
costs the wave about as much as the section called "(empty)" shows -- subtract it from each event.

    python tools/trace_summary.py gpurun_out/<tag>/team_stream_wave_trace.txt
"""
import collections
import sys

NAMES = {109: "loop top (headers, prefetch of the next read's descriptors)", 100: "descriptors -> LDS", 101: "stream (a round)",
         102: "ambiguous k-mers", 103: "epilogue return", 104: "partial list emitted", 105: "shards' lists added", 10: "epilogue entered", 0: "correction sweep", 1: "tau",
         2: "scan sweep", 3: "rank", 4: "partial sum", 5: "(empty)", 6: "publish", 7: "clear",
         # the epilogue over the touched quads (team_epilogue.hpp)
         21: "touched quads listed; quads, counts, scores in", 22: "correction (quads)", 23: "tau (quads)",
         24: "terms, candidates (quads)", 25: "rank, partial sum, publish (quads)", 26: "empty slice"}


def main():
    rows = [tuple(map(int, line.split())) for line in open(sys.argv[1])]
    reads, cur = [], []
    for code, cycles in rows:
        if code == 109 and cur:
            reads.append(cur)
            cur = []
        cur.append((code, cycles))
    reads = reads[min(50, len(reads) // 5):len(reads) - min(5, len(reads) // 10)]  # steady state (a workgroup of the product's grid places some fifty reads)
    total, count = collections.Counter(), collections.Counter()
    for r in reads:
        for code, cycles in r:
            total[code] += cycles
            count[code] += 1
    n = len(reads)
    per_read = sum(total.values()) / n
    print(f"{n} reads of one wave (one slice of each); {per_read:.0f} cycles per read with the stamps")
    for code in (109, 100, 101, 102, 104, 105, 10, 0, 1, 2, 3, 4, 5, 6, 7, 20, 21, 22, 23, 24, 25, 26, 103):
        if count[code]:
            print(f"  {NAMES[code]:62s} {total[code] / n:8.0f} cycles per read ({100 * total[code] / n / per_read:4.1f} %), "
                  f"{count[code] / n:4.2f} per read, {total[code] / count[code]:7.0f} each")


if __name__ == "__main__":
    main()
