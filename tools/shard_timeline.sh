#!/bin/bash
# Single-wave timelines of the two halves of a k-mer-space-sharded placement (diagnostic build; developer tool):
#   bash tools/shard_timeline.sh <tag> [bench args, e.g. --shard-of 8]
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-shard}; shift
OUT=$R/gpurun_out/r05/$TAG
mkdir -p $OUT
cd $R
for half in accumulate finish; do
  EPIK_AMD_LIB=$R/epik_amd/libepik_amd_ablate.so EPIK_AMD_STAMPS=1 EPIK_AMD_TRACE_HALF=$half EPIK_AMD_TRACE_FILE=$OUT/trace_$half.txt \
    timeout -k 10 300 python3 bench.py --mode kmer-shard --leaves 5000 --reads-per-step 65536 --steps 3 --warmup 1 --cpu-baseline-seconds 0 --no-extras "$@" > $OUT/bench_$half.json 2> $OUT/bench_$half.err
  echo "== $half"
  python3 tools/trace_summary.py $OUT/trace_$half.txt | tee $OUT/timeline_$half.txt
done
