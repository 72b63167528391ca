#!/bin/bash
# Round-4 experiment driver (developer tool): side-by-side timing of library builds on the large-tree workload and the
# single-wave timeline of the diagnostic build.  bash tools/r04_exp.sh <tag> [variants...]
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-exp}; shift
OUT=$R/gpurun_out/r04/$TAG
mkdir -p $OUT
cd $R
export LEAVES=${LEAVES:-5000}
echo "== side by side (LEAVES=$LEAVES): $*"
ROUNDS=${ROUNDS:-3} timeout -k 10 500 python3 tools/ablate.py "$@" 2>&1 | tee $OUT/ablate.txt
if [[ -n "$TIMELINE" ]]; then
  echo "== timeline of one wave (diagnostic build)"
  EPIK_AMD_TRACE_FILE=$OUT/wave_trace.txt ROUNDS=1 timeout -k 10 300 python3 tools/ablate.py lib=_ablate,kernel=team4,wide=2,stamps=1 > $OUT/wave_trace.log 2>&1
  python3 tools/trace_summary.py $OUT/wave_trace.txt | tee $OUT/wave_timeline.txt
fi
