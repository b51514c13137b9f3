#!/bin/bash
# The final measurement set of a round on one box: tools/measure_round.sh for every bench workload (cfg 2 with the PMC passes).
#   bash tools/measure_all.sh <tag>      ->  gpurun_out/<tag>_{bench.json,kernel_stats.csv,step_breakdown.csv,pmc_summary.json}, <tag>_<cfg>_*
TAG=$1
bash tools/measure_round.sh ${TAG} 384-grid-2x3 pmc
for pair in cfg1:384-grid-1x2 cfg3:512-grid-2x3 cfg5:384-grid-3x4 sdedit:1024-sdedit-upsample p34:384-grid-2x3-p34 mixed:384-grid-2x3-mixed g5x5:384-grid-5x5; do
  bash tools/measure_round.sh ${TAG}_${pair%%:*} ${pair##*:}
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}*_bench.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], r["value"], "steps/s", r["ms_per_step"], "ms", "gemm", r["roofline"]["frac"], "attn", r["attention_kernel"]["frac"], r["attention_kernel"]["avg_launch_us"], "board", r.get("board", {}).get("power_w_avg"), "W", r.get("board", {}).get("sclk_mhz_avg"), "MHz")
    except Exception as e:
        print(f, "FAILED", e)
PY
