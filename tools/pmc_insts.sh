#!/bin/bash
# Instruction counts of the placement kernel per ablation variant (developer tool).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_insts
mkdir -p $OUT
for ab in 0 2 10 26; do
  EPIK_AMD_LIB=$R/epik_amd/libepik_amd_ablate.so EPIK_AMD_ABLATE=$ab timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/ab$ab -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline-seconds 0 > $OUT/ab$ab.log 2>&1
  echo "== ablate=$ab (layout ${EPIK_AMD_LAYOUT:-default})"
  python3 - <<PY
import csv,glob
acc={}
for f in glob.glob("$OUT/ab$ab/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "place_reads_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
print("  ".join(f"{k.replace('SQ_','')}={sum(v)/len(v)/1e6:.0f}" for k,v in sorted(acc.items())), "(per read = value in M)")
PY
done
