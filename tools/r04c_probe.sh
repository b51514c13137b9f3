export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
export VC_PARITY_LOG=$OUT/r04c_parity.log
timeout 1500 python -m pytest tests -q -x -m gpu > $OUT/r04c_pytest.log 2>&1; tail -5 $OUT/r04c_pytest.log
for WL in 384-grid-2x3 384-grid-2x3-p34 384-grid-2x3-mixed 384-grid-1x2; do
  python bench.py --workload $WL --no-cpu-baseline --no-traffic > $OUT/r04c_${WL}.json 2> $OUT/r04c_${WL}.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/r04c_${WL}.json").read().strip().splitlines()[-1])
    print("$WL", "steps/s", r["value"], "ms/step", r["ms_per_step"], "TF/eval", r["model_tflops_per_eval"], "PF/s", r["achieved_model_tflops_per_gpu"], "gemm", r["roofline"]["frac"], r["roofline"]["avg_launch_us"], "attn", r["attention_kernel"]["frac"], r["attention_kernel"]["avg_launch_us"], r["attention_kernel"]["kernel"][:22])
except Exception as e:
    print("$WL FAILED", e)
PY
done
python tools/pack_bench.py > $OUT/r04c_pack_bench.log 2>&1; tail -14 $OUT/r04c_pack_bench.log | head -13
rocprofv3 --kernel-trace --stats -d $OUT/r04c_packprof -o pack -- python tools/pack_bench.py --iters 50 > $OUT/r04c_packprof.log 2>&1
DB=$(ls $OUT/r04c_packprof/*results.db 2>/dev/null | head -1); python tools/rocprof_summary.py $DB > $OUT/r04c_pack_kernel_stats.csv; head -8 $OUT/r04c_pack_kernel_stats.csv; rm -rf $OUT/r04c_packprof
