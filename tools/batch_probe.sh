#!/bin/bash
# Per-GPU batching probe: bench.py --per-gpu-batch {1,2,4} on the geometries whose GEMM tile counts are
# far from whole rounds of the 256 CUs;  bash tools/batch_probe.sh <tag> [workloads...]
TAG=$1; shift
WLS=${@:-"1024-sdedit-upsample 384-grid-1x2"}
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
for WL in $WLS; do
  for PB in 1 2 4; do
    python bench.py --workload $WL --per-gpu-batch $PB --no-cpu-baseline --no-traffic $BENCH_EXTRA > $OUT/${TAG}_${WL}_pb${PB}.json 2> $OUT/${TAG}_${WL}_pb${PB}.err
    python - <<PY
import json
try:
    r = json.loads(open("$OUT/${TAG}_${WL}_pb${PB}.json").read().strip().splitlines()[-1])
    print("$WL", "PB=$PB", "steps/s", r["value"], "ms/step", r["ms_per_step"], "gemm frac", r["roofline"]["frac"], "attn frac", r["attention_kernel"]["frac"])
except Exception as e:
    print("$WL PB=$PB FAILED", e)
PY
  done
done
