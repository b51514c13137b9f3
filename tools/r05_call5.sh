#!/bin/bash
# round 5, GPU call 5: the whole GPU suite (5x5 grid, reference full-width vectors, stream remainder in the auto plan), bench at 5x5 / cfg 3
mkdir -p gpurun_out
VC_PARITY_LOG=gpurun_out/r05e_parity.log python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r05e_pytest.log 2>&1
echo "pytest rc=$?" > gpurun_out/r05e_rc.txt
python bench.py --workload 384-grid-5x5 --no-cpu-baseline > gpurun_out/r05e_bench_5x5.json 2> gpurun_out/r05e_bench_5x5.err
python bench.py --workload 512-grid-2x3 --no-cpu-baseline > gpurun_out/r05e_bench_cfg3.json 2> gpurun_out/r05e_bench_cfg3.err
python bench.py > gpurun_out/r05e_bench.json 2> gpurun_out/r05e_bench.err
tail -n 25 gpurun_out/r05e_pytest.log
cat gpurun_out/r05e_rc.txt
for f in gpurun_out/r05e_bench_5x5.json gpurun_out/r05e_bench_cfg3.json gpurun_out/r05e_bench.json; do python - "$f" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], r["value"], "steps/s", r["ms_per_step"], "ms gemm", r["roofline"]["frac"], r["roofline"]["avg_launch_us"], "attn", r["attention_kernel"]["frac"], r["attention_kernel"]["avg_launch_us"], r.get("board"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
tail -5 gpurun_out/r05e_bench_5x5.err
