#!/bin/bash
# Runs tools/scratch/probe_sector plain, then under rocprofv3 --pmc (request sizes, fetch bytes).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/tools/scratch
[ -x $R/tools/scratch/probe_sector ] || hipcc -O2 --offload-arch=gfx950 -o $R/tools/scratch/probe_sector $R/tools/probe_sector.hip
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/probe_sector
mkdir -p $OUT
timeout -k 10 120 $R/tools/scratch/probe_sector > $OUT/timing.txt 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/rdreq -- $R/tools/scratch/probe_sector > $OUT/rdreq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $R/tools/scratch/probe_sector > $OUT/fetch.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("rdreq", "fetch"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % d):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print(f"{k:60s} {c:28s} last={v[-1]:.4g}")
PY
cat $OUT/timing.txt
