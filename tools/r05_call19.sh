#!/bin/bash
# round 5, GPU call 19: last check of the final code: smoke, the full-width blocks at the geometries with split / stream remainders, race screen, bench line
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05x_smoke.log 2>&1
echo "smoke rc=$?" > gpurun_out/r05x_rc.txt
python -m pytest tests/test_fullsize_gpu.py -q -k "one_plus_one and (cfg3 or sdedit or cfg1) or race or reference_itself" > gpurun_out/r05x_full.log 2>&1
echo "full rc=$?" >> gpurun_out/r05x_rc.txt
python bench.py > gpurun_out/r05x_bench.json 2> gpurun_out/r05x_bench.err
tail -n 1 gpurun_out/r05x_smoke.log; tail -n 2 gpurun_out/r05x_full.log; cat gpurun_out/r05x_rc.txt; tail -c 500 gpurun_out/r05x_bench.json
