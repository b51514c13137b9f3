#!/bin/bash
# The measurement set of a round on the GPU box (one MI355X):  bash tools/measure_round.sh <tag> [workload] [pmc]
#   <tag>_bench.json           the one JSON line of bench.py (roofline, attention_kernel, cpu_baseline inside)
#   <tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats of the same command, per kernel (tools/rocprof_summary.py)
#   <tag>_step_breakdown.csv   the trace cut into solver steps (tools/rocprof_gaps.py)
#   <tag>_pmc_summary.json     (with "pmc") four separate --pmc passes, per-kernel means (tools/pmc_summary.py)
# Copy what is to be judged from gpurun_out/ into profiles/.
TAG=$1; WL=${2:-384-grid-2x3}; PMC=$3
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
python bench.py --workload $WL $BENCH_EXTRA > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json
rm -rf $OUT/${TAG}_prof
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o ${TAG} -- python bench.py --workload $WL --no-cpu-baseline --no-traffic $BENCH_EXTRA > $OUT/${TAG}_prof.log 2>&1
DB=$(ls $OUT/${TAG}_prof/*results.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB > $OUT/${TAG}_kernel_stats.csv
python tools/rocprof_gaps.py $DB > $OUT/${TAG}_step_breakdown.csv
head -12 $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_prof        # the database stays on the box (tens of MB); the two summaries are the evidence
if [ "$PMC" = "pmc" ]; then
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $ctrs -d $OUT/${TAG}_pmc/p$i -o p$i --output-format csv -- python bench.py --workload $WL --no-cpu-baseline --no-traffic --steps 2 --warmup 1 $BENCH_EXTRA > $OUT/${TAG}_pmc_p$i.log 2>&1
  done
  python tools/pmc_summary.py $OUT/${TAG}_pmc/p*/p*_counter_collection.csv > $OUT/${TAG}_pmc_summary.json
  rm -rf $OUT/${TAG}_pmc
  head -c 1500 $OUT/${TAG}_pmc_summary.json
fi
