#!/bin/bash
# round 5, GPU call 13: no integer division per row in the qkv epilogue's pass 2: parity, phase times, interleaved A/B
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -q -k "qkv or head_permuted or persistent" > gpurun_out/r05k_ops.log 2>&1
echo "ops rc=$?" > gpurun_out/r05k_rc.txt
python -m pytest tests/test_handle_gpu.py tests/test_model_gpu.py -q > gpurun_out/r05k_model.log 2>&1
echo "model rc=$?" >> gpurun_out/r05k_rc.txt
python -m pytest tests/test_fullsize_gpu.py -q -k "batch_of_two or one_plus_one and cfg2 or fused_sampler and (cfg1 or cfg2)" > gpurun_out/r05k_full.log 2>&1
echo "full rc=$?" >> gpurun_out/r05k_rc.txt
VC_HIP_LIB=visualcloze_amd/lib/libvcloze_hip_dbg.so python tools/qkv_epilogue_phases.py > gpurun_out/r05k_qkv_phases.log 2>&1
python tools/step_ab.py main=nodiv divs=divs --rounds 7 > gpurun_out/r05k_ab_cfg2.log 2>&1
python tools/step_ab.py main=nodiv divs=divs --rounds 3 --workload 384-grid-3x4 > gpurun_out/r05k_ab_cfg5.log 2>&1
tail -n 3 gpurun_out/r05k_ops.log gpurun_out/r05k_model.log gpurun_out/r05k_full.log; cat gpurun_out/r05k_rc.txt
grep -v amdgpu.ids gpurun_out/r05k_qkv_phases.log | tail -7; grep -hv amdgpu.ids gpurun_out/r05k_ab_cfg2.log gpurun_out/r05k_ab_cfg5.log | cut -c1-200
