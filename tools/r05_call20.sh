#!/bin/bash
# round 5, GPU call 20: the full-depth tests on the final code (the one family not re-run after the last two changes)
mkdir -p gpurun_out
timeout 170 python -m pytest tests/test_fullsize_gpu.py -q -x -k "full_depth or full_model" > gpurun_out/r05x_fulldepth.log 2>&1
echo "fulldepth rc=$?" > gpurun_out/r05x_rc2.txt
tail -n 3 gpurun_out/r05x_fulldepth.log; cat gpurun_out/r05x_rc2.txt
