#!/bin/bash
# Side by side of two library builds on the bench workloads (developer tool): bash tools/ab.sh  (libepik_amd_base.so against libepik_amd.so)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for w in "LEAVES=500" "LEAVES=1000" "LEAVES=1500" "LEAVES=5000" "LEAVES=5000 CLADES=1"; do
  echo "== $w"
  env $w ROUNDS=${ROUNDS:-3} timeout -k 10 300 python3 tools/ablate.py lib=_base,layout=${LAYOUT:-paired} layout=${LAYOUT:-paired} 2>&1 | grep "reads/s"
done
