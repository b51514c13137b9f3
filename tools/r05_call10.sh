#!/bin/bash
# round 5, GPU call 10: the whole GPU suite on the final code, then the final measurement set on the same box
mkdir -p gpurun_out
VC_PARITY_LOG=gpurun_out/r05z_parity.log python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r05z_pytest.log 2>&1
echo "pytest rc=$?" > gpurun_out/r05z_rc.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05z_smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r05z_rc.txt
bash tools/measure_all.sh r05z > gpurun_out/r05z_measure_all.log 2>&1
tail -n 14 gpurun_out/r05z_pytest.log; cat gpurun_out/r05z_rc.txt; tail -n 2 gpurun_out/r05z_smoke.log; tail -n 9 gpurun_out/r05z_measure_all.log
