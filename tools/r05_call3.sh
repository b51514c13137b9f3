#!/bin/bash
# round 5, GPU call 3: last-arriver combine on L2-local sync (placement probed), world-2 real-engine tests, A/B vs the merge kernel
mkdir -p gpurun_out
python -m pytest tests/test_parallel_gpu.py -q -x > gpurun_out/r05c_world2.log 2>&1
echo "world2 rc=$?" > gpurun_out/r05c_rc.txt
python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_handle_gpu.py tests/test_model_gpu.py -q -k "attention or race or fused or handle or route or batch" > gpurun_out/r05c_attn.log 2>&1
echo "attn rc=$?" >> gpurun_out/r05c_rc.txt
python tools/step_ab.py main=l2local mergek=mergek --attn --rounds 7 > gpurun_out/r05c_ab_cfg2.log 2>&1
python tools/step_ab.py main=l2local mergek=mergek --attn --rounds 3 --workload 512-grid-2x3 > gpurun_out/r05c_ab_cfg3.log 2>&1
python tools/step_ab.py main=l2local mergek=mergek --attn --rounds 3 --workload 384-grid-3x4 > gpurun_out/r05c_ab_cfg5.log 2>&1
python bench.py > gpurun_out/r05c_bench.json 2> gpurun_out/r05c_bench.err
tail -n 3 gpurun_out/r05c_world2.log gpurun_out/r05c_attn.log
cat gpurun_out/r05c_rc.txt; grep -hv amdgpu.ids gpurun_out/r05c_ab_cfg2.log gpurun_out/r05c_ab_cfg3.log gpurun_out/r05c_ab_cfg5.log
tail -c 700 gpurun_out/r05c_bench.json
