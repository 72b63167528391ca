#!/bin/bash
# rocprofv3 kernel trace of one bench.py run, per-kernel average durations (developer tool):
#   bash tools/kt.sh <name> [bench args]        EPIK_AMD_LIB picks the library
R=${GRAFT_REPO_ROOT:-$PWD}
name=$1; shift
OUT=$R/gpurun_out/r05/kt_$name
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 --no-extras "$@" > $OUT/bench.json 2> $OUT/bench.err || echo "trace $name failed"
cp $(ls $OUT/*/*kernel_stats.csv | head -1) $R/gpurun_out/r05/kernel_stats_$name.csv 2>/dev/null
python3 - <<PY
import csv, json
print("== $name", round(json.load(open("$OUT/bench.json"))["value"] / 1e6, 2), "M reads/s")
for r in csv.DictReader(open("$R/gpurun_out/r05/kernel_stats_$name.csv")):
    if float(r["Percentage"]) > 0.5:
        print("   %-90s calls %4s avg %9.1f us  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
