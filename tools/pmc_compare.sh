#!/bin/bash
# SQ counters of the team kernels for several builds of the library, one rocprofv3 --pmc pass each (developer tool):
#   LIBS="_r03 ''" bash tools/pmc_compare.sh [bench args]      (suffixes of epik_amd/libepik_amd<suffix>.so)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_compare
mkdir -p $OUT
for suffix in ${LIBS:-_r03 .}; do
  [[ $suffix == . ]] && suffix=""
  name=lib${suffix:-_new}
  rm -rf $OUT/$name
  EPIK_AMD_LIB=$R/epik_amd/libepik_amd$suffix.so timeout -k 10 300 rocprofv3 --pmc ${COUNTERS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT} --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline-seconds 0 --no-extras --leaves ${LEAVES:-5000} "$@" > $OUT/$name.log 2>&1
  echo "== $name"
  python3 - <<PY
import csv,glob
acc={}
for f in glob.glob("$OUT/$name/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("<")[0].split("(")[0].replace("void epik_amd::", "")
        acc.setdefault((k, r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
for kern in sorted({k[0] for k in acc}):
    if "team" not in kern and "place" not in kern: continue
    print(" ", kern, "  ".join(f"{k[1].replace('SQ_','')}={sum(v)/len(v)/1e6:.1f}" for k,v in sorted(acc.items()) if k[0]==kern), "(millions per launch)")
PY
done
