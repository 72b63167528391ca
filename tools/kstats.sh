#!/bin/bash
# Per-kernel times of one bench.py configuration: tools/kstats.sh <tag> <bench.py arguments...>
# Writes gpurun_out/r03/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and the bench line beside it.
set -e
tag=$1; shift
out=$(pwd)/gpurun_out/r03
mkdir -p "$out/prof_$tag"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/prof_$tag" -o "$tag" --output-format csv -- python3 bench.py "$@" --no-extras --cpu-baseline-seconds 0 > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.log"
f=$(find "$out/prof_$tag" -name "*kernel_stats.csv" | head -1)
cp "$f" "$out/${tag}_kernel_stats.csv"
cut -d, -f1-4,6-8 "$out/${tag}_kernel_stats.csv" | head -12
rm -rf "$out/prof_$tag"
