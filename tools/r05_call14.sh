#!/bin/bash
# round 5, GPU call 14: DPP adds instead of ds_bpermute in qknorm_rope8's row reduction: parity (bit-identical by construction), A/B
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_golden_ops_gpu.py -q -k "qk or head_permuted or rope or prescaled or query" > gpurun_out/r05l_ops.log 2>&1
echo "ops rc=$?" > gpurun_out/r05l_rc.txt
python tools/step_ab.py main=dpp bperm=bperm --rounds 7 > gpurun_out/r05l_ab_cfg2.log 2>&1
python tools/step_ab.py main=dpp bperm=bperm --rounds 3 --workload 384-grid-1x2 > gpurun_out/r05l_ab_cfg1.log 2>&1
tail -n 3 gpurun_out/r05l_ops.log; cat gpurun_out/r05l_rc.txt; grep -hv amdgpu.ids gpurun_out/r05l_ab_cfg2.log gpurun_out/r05l_ab_cfg1.log | cut -c1-200
