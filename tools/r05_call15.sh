#!/bin/bash
# round 5, GPU call 15: the whole GPU suite on the FINAL code, then the final measurement set on the same box
mkdir -p gpurun_out
VC_PARITY_LOG=gpurun_out/r05y_parity.log python -m pytest tests -q -m gpu > gpurun_out/r05y_pytest.log 2>&1
echo "pytest rc=$?" > gpurun_out/r05y_rc.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05y_smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r05y_rc.txt
bash tools/measure_all.sh r05y > gpurun_out/r05y_measure_all.log 2>&1
tail -n 4 gpurun_out/r05y_pytest.log; cat gpurun_out/r05y_rc.txt; tail -n 2 gpurun_out/r05y_smoke.log; tail -n 9 gpurun_out/r05y_measure_all.log
