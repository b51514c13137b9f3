#!/bin/bash
# round 5, GPU call 9: qknorm_rope8 written for instruction count (explicit FMAs, v_rsq_f32): parity + interleaved A/B
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_golden_ops_gpu.py tests/test_model_gpu.py tests/test_handle_gpu.py -q > gpurun_out/r05i_tests.log 2>&1
echo "tests rc=$?" > gpurun_out/r05i_rc.txt
python -m pytest tests/test_fullsize_gpu.py -q -k "one_plus_one and (cfg1 or cfg2) or reference_itself or trajectory_vs_oracle and cfg2" > gpurun_out/r05i_full.log 2>&1
echo "full rc=$?" >> gpurun_out/r05i_rc.txt
python tools/step_ab.py main=leannorm oldnorm=oldnorm --rounds 7 > gpurun_out/r05i_ab_cfg2.log 2>&1
python tools/step_ab.py main=leannorm oldnorm=oldnorm --rounds 3 --workload 384-grid-3x4 > gpurun_out/r05i_ab_cfg5.log 2>&1
python tools/step_ab.py main=leannorm oldnorm=oldnorm --rounds 3 --workload 384-grid-1x2 > gpurun_out/r05i_ab_cfg1.log 2>&1
tail -n 4 gpurun_out/r05i_tests.log gpurun_out/r05i_full.log
cat gpurun_out/r05i_rc.txt; grep -hv amdgpu.ids gpurun_out/r05i_ab_cfg2.log gpurun_out/r05i_ab_cfg5.log gpurun_out/r05i_ab_cfg1.log | cut -c1-200
