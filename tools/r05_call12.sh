#!/bin/bash
# round 5, GPU call 12: staging form of the GEMM epilogue decided outside the unrolled loops: parity, phase times, interleaved A/B
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -q -k "gemm" > gpurun_out/r05j_ops.log 2>&1
echo "ops rc=$?" > gpurun_out/r05j_rc.txt
python -m pytest tests/test_handle_gpu.py tests/test_model_gpu.py tests/test_vae_gpu.py tests/test_text_gpu.py -q > gpurun_out/r05j_model.log 2>&1
echo "model rc=$?" >> gpurun_out/r05j_rc.txt
VC_HIP_LIB=visualcloze_amd/lib/libvcloze_hip_dbg.so python tools/qkv_epilogue_phases.py > gpurun_out/r05j_qkv_phases.log 2>&1
python tools/step_ab.py main=hoisted branchy=branchy --rounds 7 > gpurun_out/r05j_ab_cfg2.log 2>&1
python tools/step_ab.py main=hoisted branchy=branchy --rounds 3 --workload 512-grid-2x3 > gpurun_out/r05j_ab_cfg3.log 2>&1
python tools/step_ab.py main=hoisted branchy=branchy --rounds 3 --workload 384-grid-1x2 > gpurun_out/r05j_ab_cfg1.log 2>&1
tail -n 3 gpurun_out/r05j_ops.log gpurun_out/r05j_model.log; cat gpurun_out/r05j_rc.txt
grep -v amdgpu.ids gpurun_out/r05j_qkv_phases.log | tail -7; grep -hv amdgpu.ids gpurun_out/r05j_ab_cfg2.log gpurun_out/r05j_ab_cfg3.log gpurun_out/r05j_ab_cfg1.log | cut -c1-200
