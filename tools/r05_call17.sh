#!/bin/bash
# round 5, GPU call 17: one split-K scratch per handle (bound) instead of one per cached workspace: handle / C-ABI / sampler parity
mkdir -p gpurun_out
python -m pytest tests/test_handle_gpu.py tests/test_c_abi.py tests/test_model_gpu.py tests/test_parallel_gpu.py -q > gpurun_out/r05m_tests.log 2>&1
echo "tests rc=$?" > gpurun_out/r05m_rc.txt
python -m pytest tests/test_fullsize_gpu.py -q -k "fused_sampler or trajectory_vs_oracle and sdedit or batch_of_two" > gpurun_out/r05m_full.log 2>&1
echo "full rc=$?" >> gpurun_out/r05m_rc.txt
python bench.py --workload 1024-sdedit-upsample --no-cpu-baseline --no-traffic > gpurun_out/r05m_bench_sdedit.json 2> gpurun_out/r05m_bench_sdedit.err
tail -n 3 gpurun_out/r05m_tests.log gpurun_out/r05m_full.log; cat gpurun_out/r05m_rc.txt; tail -c 300 gpurun_out/r05m_bench_sdedit.json
