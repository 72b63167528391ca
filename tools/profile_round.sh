#!/bin/bash
# Everything the round's measurements under profiles/ come from, in one GPU call:
#   bash tools/profile_round.sh r02_a        (on the GPU box; results under gpurun_out/<tag>/)
# bench lines, rocprofv3 --kernel-trace --stats summaries of the same commands, and the PMC passes
# (separate rocprofv3 --pmc runs, tools/pmc_passes.sh) for the headline workload, the HBM-resident
# database (k = 11) and the large tree (N = 9 999, team kernels); SQ counters, per-phase instruction counts and a
# single-wave timeline of the team kernels; the tree-size sweep.
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r05}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
log() { echo "[profile_round] $*"; }

# PART=1: bench lines, kernel traces, driver and shard rates; PART=3: PMC passes; PART=2: SQ counters, instruction
# counts, timelines; PART=4: the tree-size sweep (a GPU call is limited to 20 minutes: four calls); default: all.
PART=${PART:-1234}
if [[ $PART == *1* ]]; then
log "full default bench line"
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || log "bench failed"

trace() { # name args...
  local name=$1; shift
  log "kernel trace: $name"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$name -- \
      python3 $R/bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err \
      || log "trace $name failed"
  cp $(ls $OUT/kt_$name/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_$name.csv 2>/dev/null
}
trace headline
trace k11 --kmer-size 11
trace n9999 --leaves 5000
trace n9999_clades --leaves 5000 --clades
trace n2999 --leaves 1500
# configs[3] at its stated size: ~1 G postings, the sparse form of the descriptor
trace amino_k7 --states amino --kmer-size 7 --read-length 300 --p-present 0.0133
trace kmer_shard_n9999 --mode kmer-shard --leaves 5000 --reads-per-step 65536
trace kmer_shard_n9999_256k --mode kmer-shard --leaves 5000 --reads-per-step 262144
trace kmer_shard_n9999_0of8 --mode kmer-shard --leaves 5000 --reads-per-step 65536 --shard-of 8
# (the presence filter of the protein database in 64-bit words: what round 3 measured; the default is the packed one)
EPIK_AMD_FILTER=wide trace amino_k7_wide_filter --states amino --kmer-size 7 --read-length 300 --p-present 0.0133
log "end to end through the native driver, 1 M reads"
(cd $R && python3 tools/e2e_bench.py --reads 1000000 --batch-size 2000 --jobs 1 4 16 > $OUT/e2e_driver.txt 2>&1; python3 tools/e2e_bench.py --reads 1000000 --batch-size 2000 --jobs 16 --devices 0,0 >> $OUT/e2e_driver.txt 2>&1)
log "epik_amd_placer_place_sharded (the --db-shard path), N = 9999"
(cd $R && python3 tools/shard_rate.py > $OUT/shard_rate.txt 2>&1)

fi
if [[ $PART == *3* ]]; then
for cfg in "headline:" "k11:--kmer-size 11" "n9999:--leaves 5000" "n9999_clades:--leaves 5000 --clades" "amino_k7:--states amino --kmer-size 7 --read-length 300 --p-present 0.0133"; do
  name=${cfg%%:*}; args=${cfg#*:}
  log "PMC passes: $name"
  (cd $R && BENCH_ARGS="$args" bash tools/pmc_passes.sh ${TAG}_$name > $OUT/pmc_$name.log 2>&1)
  cp $R/gpurun_out/pmc_${TAG}_$name/summary.txt $OUT/pmc_summary_$name.txt 2>/dev/null
done
fi
if [[ $PART == *2* ]]; then
for cfg in "headline:" "n9999_team:--leaves 5000"; do
  name=${cfg%%:*}; args=${cfg#*:}
  log "SQ / TCP counter passes: $name"
  (cd $R && BENCH_ARGS="$args" bash tools/pmc_sq.sh ${TAG}_sq_$name > $OUT/sq_$name.log 2>&1)
  cp $R/gpurun_out/pmc_${TAG}_sq_$name/summary.txt $OUT/sq_counters_$name.txt 2>/dev/null
done
# (how busy the vector ALUs are on the other two workloads: the instruction counters only)
for cfg in "k11:--kmer-size 11" "amino_k7:--states amino --kmer-size 7 --read-length 300 --p-present 0.0133"; do
  name=${cfg%%:*}; args=${cfg#*:}
  log "SQ counter passes (sq1, sq2): $name"
  (cd $R && BENCH_ARGS="$args" SQ_PASSES="sq1 sq2" SQ_TIMEOUT=500 bash tools/pmc_sq.sh ${TAG}_sq_$name > $OUT/sq_$name.log 2>&1)
  cp $R/gpurun_out/pmc_${TAG}_sq_$name/summary.txt $OUT/sq_counters_$name.txt 2>/dev/null
done
log "instruction counts of the team kernels by phase (diagnostic build)"
(cd $R && make -C epik_amd/csrc ablate > /dev/null 2>&1 && bash tools/pmc_insts_team.sh > $OUT/team_instruction_counts.txt 2>&1)
log "timelines of one wave of the two halves of the k-mer-space shard (diagnostic build)"
(cd $R && bash tools/shard_timeline.sh ${TAG}_shard_g1 > $OUT/shard_halves_wave_timeline.txt 2>&1; bash tools/shard_timeline.sh ${TAG}_shard_0of8 --shard-of 8 > $OUT/shard_halves_0of8_wave_timeline.txt 2>&1)
log "timeline of one wave of the streaming kernel on the clade workload (diagnostic build)"
(cd $R && CLADES=1 EPIK_AMD_TRACE_FILE=$OUT/team_stream_wave_trace_clades.txt LEAVES=5000 ROUNDS=1 python3 tools/ablate.py lib=_ablate,kernel=team4,wide=2,stamps=1 > $OUT/team_stream_wave_trace_clades.log 2>&1 && python3 tools/trace_summary.py $OUT/team_stream_wave_trace_clades.txt > $OUT/team_stream_wave_timeline_clades.txt)
log "timeline of one wave of the streaming kernel (diagnostic build)"
(cd $R && EPIK_AMD_TRACE_FILE=$OUT/team_stream_wave_trace.txt LEAVES=5000 ROUNDS=1 python3 tools/ablate.py lib=_ablate,kernel=team4,wide=2,stamps=1 > $OUT/team_stream_wave_trace.log 2>&1 && python3 tools/trace_summary.py $OUT/team_stream_wave_trace.txt > $OUT/team_stream_wave_timeline.txt)
fi
if [[ $PART == *4* ]]; then
log "tree sizes, kernels, passes"
(cd $R && bash tools/sweep_tree_sizes.sh > $OUT/sweep_tree_sizes_passes.txt 2>&1)
fi
log done
ls $OUT
