#!/bin/bash
# round 5, GPU call 18: pass 2 of the GEMM epilogue without a division / 64-bit multiply per iteration: parity (bit-identical), A/B
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -q -k "gemm" > gpurun_out/r05n_ops.log 2>&1
echo "ops rc=$?" > gpurun_out/r05n_rc.txt
python -m pytest tests/test_handle_gpu.py tests/test_vae_gpu.py tests/test_text_gpu.py -q > gpurun_out/r05n_model.log 2>&1
echo "model rc=$?" >> gpurun_out/r05n_rc.txt
python tools/step_ab.py main=stepped idxdiv=idxdiv --rounds 6 > gpurun_out/r05n_ab_cfg2.log 2>&1
tail -n 2 gpurun_out/r05n_ops.log gpurun_out/r05n_model.log; cat gpurun_out/r05n_rc.txt; grep -hv amdgpu.ids gpurun_out/r05n_ab_cfg2.log | cut -c1-200
