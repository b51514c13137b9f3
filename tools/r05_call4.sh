#!/bin/bash
# round 5, GPU call 4: stream form of the split-K remainder - parity, then interleaved A/B at cfg 3 / cfg 5 / p34
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -q -k "stream or splitk or attention" > gpurun_out/r05d_ops.log 2>&1
echo "ops rc=$?" > gpurun_out/r05d_rc.txt
python -m pytest tests/test_fullsize_gpu.py -q -k "race or properties" > gpurun_out/r05d_full.log 2>&1
echo "full rc=$?" >> gpurun_out/r05d_rc.txt
P=$((1<<22)); A=$(( (1<<22) | (1<<23) ))
python tools/step_ab.py main=cut main=stream main=streamall --opt stream:tile_cfg=$P --opt streamall:tile_cfg=$A --rounds 4 --workload 512-grid-2x3 > gpurun_out/r05d_ab_cfg3.log 2>&1
python tools/step_ab.py main=auto main=stream main=streamall --opt stream:tile_cfg=$P --opt streamall:tile_cfg=$A --rounds 4 --workload 384-grid-3x4 > gpurun_out/r05d_ab_cfg5.log 2>&1
python tools/step_ab.py main=auto main=stream --opt stream:tile_cfg=$P --rounds 4 --workload 384-grid-2x3-p34 > gpurun_out/r05d_ab_p34.log 2>&1
tail -n 3 gpurun_out/r05d_ops.log gpurun_out/r05d_full.log
cat gpurun_out/r05d_rc.txt; grep -hv amdgpu.ids gpurun_out/r05d_ab_cfg3.log gpurun_out/r05d_ab_cfg5.log gpurun_out/r05d_ab_p34.log
