#!/bin/bash
# SQ / TCP / TLB counter passes on a short bench.py run (developer tool).
# Usage on the GPU box: EPIK_AMD_LAYOUT=... bash tools/pmc_sq.sh <tag>
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-sq}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
# SQ_PASSES="sq1 sq2": only those passes (default: all)
pass() { local name=$1; shift
  if [[ -n "$SQ_PASSES" && " $SQ_PASSES " != *" $name "* ]]; then return; fi
  timeout -k 10 ${SQ_TIMEOUT:-300} rocprofv3 --pmc "$@" --output-format csv -d $OUT/bench_$name -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0 --no-extras $BENCH_ARGS > $OUT/bench_$name.log 2>&1 || echo "pass $name failed"
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT
pass sq3 SQ_WAVES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH GRBM_GUI_ACTIVE
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass tcp2 TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
