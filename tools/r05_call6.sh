#!/bin/bash
# round 5, GPU call 6: the final measurement set on one box (all workloads: bench line + rocprofv3 kernel stats + step breakdown; cfg 2 with PMC)
bash tools/measure_all.sh r05w > gpurun_out/r05w_measure_all.log 2>&1
tail -n 12 gpurun_out/r05w_measure_all.log
