#!/bin/bash
mkdir -p gpurun_out
VC_HIP_LIB=visualcloze_amd/lib/libvcloze_hip_dbg.so python tools/qkv_epilogue_phases.py > gpurun_out/r05g_qkv_phases.log 2>&1
cat gpurun_out/r05g_qkv_phases.log
