#!/bin/bash
# round 5, GPU call 8: balanced head / V layout of the qkv tiles (V accumulators with swapped MFMA operands, pipelined norm loop):
# parity, the epilogue's phase times, A/B against a build of the previous layout (two processes, same box, alternating twice)
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -q -k "qkv or head_permuted or prescaled or persistent" > gpurun_out/r05h_ops.log 2>&1
echo "ops rc=$?" > gpurun_out/r05h_rc.txt
python -m pytest tests/test_handle_gpu.py tests/test_model_gpu.py tests/test_golden_ops_gpu.py -q > gpurun_out/r05h_model.log 2>&1
echo "model rc=$?" >> gpurun_out/r05h_rc.txt
python -m pytest tests/test_fullsize_gpu.py -q -k "one_plus_one and (cfg1 or cfg2 or p34) or reference_itself or fused_sampler" > gpurun_out/r05h_full.log 2>&1
echo "full rc=$?" >> gpurun_out/r05h_rc.txt
VC_HIP_LIB=visualcloze_amd/lib/libvcloze_hip_dbg.so python tools/qkv_epilogue_phases.py > gpurun_out/r05h_qkv_phases.log 2>&1
for i in 1 2; do
  VC_HIP_LIB=visualcloze_amd/lib/libvcloze_hip_lay1.so VC_QKV_LAYOUT=1 python tools/step_ab.py main=lay1 --rounds 4 --attn >> gpurun_out/r05h_ab_cfg2.log 2>&1
  python tools/step_ab.py main=balanced --rounds 4 --attn >> gpurun_out/r05h_ab_cfg2.log 2>&1
done
VC_HIP_LIB=visualcloze_amd/lib/libvcloze_hip_lay1.so VC_QKV_LAYOUT=1 python tools/step_ab.py main=lay1 --rounds 3 --workload 512-grid-2x3 >> gpurun_out/r05h_ab_cfg3.log 2>&1
python tools/step_ab.py main=balanced --rounds 3 --workload 512-grid-2x3 >> gpurun_out/r05h_ab_cfg3.log 2>&1
VC_HIP_LIB=visualcloze_amd/lib/libvcloze_hip_lay1.so VC_QKV_LAYOUT=1 python tools/step_ab.py main=lay1 --rounds 3 --workload 384-grid-1x2 >> gpurun_out/r05h_ab_cfg1.log 2>&1
python tools/step_ab.py main=balanced --rounds 3 --workload 384-grid-1x2 >> gpurun_out/r05h_ab_cfg1.log 2>&1
tail -n 3 gpurun_out/r05h_ops.log gpurun_out/r05h_model.log gpurun_out/r05h_full.log
cat gpurun_out/r05h_rc.txt; grep -v amdgpu.ids gpurun_out/r05h_qkv_phases.log | tail -14; grep -hv amdgpu.ids gpurun_out/r05h_ab_cfg2.log gpurun_out/r05h_ab_cfg3.log gpurun_out/r05h_ab_cfg1.log
