export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm" > $OUT/r04b_pytest_gemm.log 2>&1; tail -3 $OUT/r04b_pytest_gemm.log
timeout 900 python -m pytest tests/test_handle_gpu.py tests/test_model_gpu.py -q -x > $OUT/r04b_pytest_handle.log 2>&1; tail -3 $OUT/r04b_pytest_handle.log
for WL in 1024-sdedit-upsample 384-grid-1x2; do
 for X in "" "--no-splitk"; do
  python bench.py --workload $WL --no-cpu-baseline --no-traffic $X > $OUT/r04b_${WL}${X}.json 2> $OUT/r04b_${WL}${X}.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/r04b_${WL}${X}.json").read().strip().splitlines()[-1])
    print("$WL", "$X", "steps/s", r["value"], "ms/step", r["ms_per_step"], "gemm frac", r["roofline"]["frac"], r["roofline"]["avg_launch_us"], "attn frac", r["attention_kernel"]["frac"])
except Exception as e:
    print("$WL $X FAILED", e)
PY
 done
done
python bench.py --workload 1024-sdedit-upsample --per-gpu-batch 2 --no-cpu-baseline --no-traffic > $OUT/r04b_sdedit_pb2.json 2>$OUT/r04b_sdedit_pb2.err; python -c "
import json; r=json.loads(open('$OUT/r04b_sdedit_pb2.json').read().strip().splitlines()[-1]); print('sdedit PB2', r['value'], r['ms_per_step'], r['roofline']['frac'])"
