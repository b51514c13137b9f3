#!/bin/bash
# round 5, GPU call 1: query norm in the qkv GEMM epilogue - parity, then interleaved A/B against the attention-prologue route
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -q -x -k "head_permuted or prescaled or persistent_grouped or qkv_epilogue or in_kernel_query" > gpurun_out/r05a_ops.log 2>&1
echo "ops rc=$?" > gpurun_out/r05a_rc.txt
python -m pytest tests -q -m gpu --deselect tests/test_ops_gpu.py -x > gpurun_out/r05a_rest.log 2>&1
echo "rest rc=$?" >> gpurun_out/r05a_rc.txt
python -m pytest tests/test_ops_gpu.py -q -m gpu > gpurun_out/r05a_ops_all.log 2>&1
echo "ops_all rc=$?" >> gpurun_out/r05a_rc.txt
python tools/step_ab.py main=prologue main=epilogue --opt prologue:fuse_qnorm=1 --opt epilogue:fuse_qnorm=2 --attn --rounds 5 > gpurun_out/r05a_qnorm_ab_cfg2.log 2>&1
python tools/step_ab.py main=prologue main=epilogue --opt prologue:fuse_qnorm=1 --opt epilogue:fuse_qnorm=2 --attn --rounds 3 --workload 512-grid-2x3 > gpurun_out/r05a_qnorm_ab_cfg3.log 2>&1
python bench.py > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err
tail -3 gpurun_out/r05a_ops.log gpurun_out/r05a_rest.log gpurun_out/r05a_ops_all.log
cat gpurun_out/r05a_rc.txt gpurun_out/r05a_qnorm_ab_cfg2.log gpurun_out/r05a_qnorm_ab_cfg3.log
tail -c 1500 gpurun_out/r05a_bench.json
