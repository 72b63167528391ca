#!/bin/bash
# Collects TCC counters in separate rocprofv3 --pmc passes (4 TCC slots per pass) for
# (a) the FETCH_SIZE calibration streams and (b) one short bench.py run.
# Usage on the GPU box: bash tools/pmc_passes.sh <tag>
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r01}
mkdir -p $R/tools/scratch
[ -x $R/tools/scratch/calib_fetch ] || hipcc -O2 --offload-arch=gfx950 -o $R/tools/scratch/calib_fetch $R/tools/calib_fetch.hip
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
pass() { # name counters...
  local name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/calib_$name -- $R/tools/scratch/calib_fetch > $OUT/calib_$name.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/bench_$name -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0 --no-extras $BENCH_ARGS > $OUT/bench_$name.log 2>&1
}
pass fetch FETCH_SIZE
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass hit TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass dram TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCP_TCC_READ_REQ_sum
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
