#!/bin/bash
# reads/s against tree size, kernel and number of passes (what db_image.cpp's kernel threshold and
# choose_team are set from): bash tools/sweep_tree_sizes.sh > gpurun_out/sweep.txt   (on the GPU box)
# SWEEP=short: the sizes around the thresholds only.
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/sweep
run() { # tag env... -- args
  tag=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python3 $R/bench.py --steps 5 --warmup 1 --cpu-baseline-seconds 0 --no-extras "$@" > $R/gpurun_out/sweep/$tag.json 2> $R/gpurun_out/sweep/$tag.err || echo "FAILED $tag"
  python3 - <<PY
import json
try:
    d=json.loads(open("$R/gpurun_out/sweep/$tag.json").read().strip().splitlines()[-1])
    print("$tag", round(d["value"]/1e6,2), "M reads/s", round(d["ms_per_step"],3), "ms", d["config"]["launch"], "frac", round(d["roofline"]["frac"],3), d["roofline"]["kernel"])
except Exception as e:
    print("$tag", "no result", e)
PY
}
if [ "$SWEEP" != "short" ]; then
run n999_default --
run n1303_default -- --leaves 652
fi
# round 4: one wavefront per read against two and four slices per pass, N = 1 499 .. 4 999 (and what create() picks)
for leaves in 750 1000 1250 1500 1750 2000 2500; do
  n=$((2*leaves-1))
  run n${n}_default -- --leaves $leaves
  run n${n}_wave EPIK_AMD_KERNEL=wave -- --leaves $leaves
  run n${n}_team2 EPIK_AMD_KERNEL=team2 -- --leaves $leaves
  run n${n}_team4 EPIK_AMD_KERNEL=team4 -- --leaves $leaves
done
for leaves in 3000 3750; do
  n=$((2*leaves-1))
  run n${n}_default -- --leaves $leaves
  run n${n}_team2 EPIK_AMD_KERNEL=team2 -- --leaves $leaves
done
run n9999_default -- --leaves 5000
run n9999_team2 EPIK_AMD_KERNEL=team2 -- --leaves 5000
run n9999_team4_dense_epilogue EPIK_AMD_KERNEL=team4 EPIK_AMD_TEAM_SPARSE=0 -- --leaves 5000
run n9999_team4_one_kernel EPIK_AMD_KERNEL=team4 EPIK_AMD_TEAM_FRONT=0 -- --leaves 5000
run n9999_team8 EPIK_AMD_KERNEL=team8 -- --leaves 5000 --reads-per-step 500000
run n9999_team4x2 EPIK_AMD_KERNEL=team4x2 -- --leaves 5000 --reads-per-step 500000
if [ "$SWEEP" != "short" ]; then
for p in 1 2; do run n14999_team4x$p EPIK_AMD_KERNEL=team4x$p -- --leaves 7500 --reads-per-step 200000; done
for p in 1 2 3; do run n19999_team4x$p EPIK_AMD_KERNEL=team4x$p -- --leaves 10000 --reads-per-step 200000; done
for p in 2 3 4; do run n29999_team4x$p EPIK_AMD_KERNEL=team4x$p -- --leaves 15000 --reads-per-step 100000; done
run n49999_default -- --leaves 25000 --reads-per-step 100000
fi
