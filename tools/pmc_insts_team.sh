#!/bin/bash
# Instruction counts of the team kernels (front, streaming) per ablation variant (developer tool):
# 0 everything, 2 no slice epilogue, 6 no epilogue and no merge, 8 nothing streamed, 14 neither.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_insts_team
mkdir -p $OUT
for ab in ${VARIANTS:-0 2 6 8 14}; do
  EPIK_AMD_LIB=$R/epik_amd/libepik_amd_ablate.so EPIK_AMD_ABLATE=$ab timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/ab$ab -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline-seconds 0 --no-extras --leaves ${LEAVES:-5000} > $OUT/ab$ab.log 2>&1
  echo "== ablate=$ab"
  python3 - <<PY
import csv,glob
acc={}
for f in glob.glob("$OUT/ab$ab/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        for kern in ("team_front_kernel", "team_stream_kernel", "team_place_kernel"):
            if kern in r["Kernel_Name"]:
                acc.setdefault((kern, r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
for kern in ("team_front_kernel", "team_stream_kernel", "team_place_kernel"):
    print(" ", kern, "  ".join(f"{k[1].replace('SQ_','')}={sum(v)/len(v)/1e6:.0f}" for k,v in sorted(acc.items()) if k[0]==kern), "(per read = value in M)")
PY
done
