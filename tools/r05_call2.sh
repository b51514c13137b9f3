#!/bin/bash
# round 5, GPU call 2: query norm in the qkv GEMM epilogue + last-arriver tail combine - full parity, then interleaved A/Bs
mkdir -p gpurun_out
( ls -la /sys/class/drm/ ; for c in /sys/class/drm/card[0-9]*; do echo "== $c -> $(readlink -f $c/device)"; ls $c/device/hwmon/*/ 2>/dev/null | tr '\n' ' '; echo;
  for f in $c/device/hwmon/*/power1_average $c/device/hwmon/*/power1_input $c/device/hwmon/*/power1_cap $c/device/hwmon/*/freq1_input $c/device/pp_dpm_sclk; do [ -e $f ] && echo "$f: $(cat $f 2>&1 | tr '\n' ' ')"; done; done;
  python -c "import torch; p=torch.cuda.get_device_properties(0); print({k: getattr(p,k) for k in dir(p) if 'pci' in k})";
  rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40 ) > gpurun_out/r05b_sysfs_probe.log 2>&1
python -m pytest tests -q -m gpu > gpurun_out/r05b_pytest.log 2>&1
echo "pytest rc=$?" > gpurun_out/r05b_rc.txt
python tools/step_ab.py main=inmerge mergek=mergek mergek=prologue --opt prologue:fuse_qnorm=1 --attn --rounds 5 > gpurun_out/r05b_ab_cfg2.log 2>&1
python tools/step_ab.py main=inmerge mergek=mergek mergek=prologue --opt prologue:fuse_qnorm=1 --attn --rounds 3 --workload 512-grid-2x3 > gpurun_out/r05b_ab_cfg3.log 2>&1
python tools/step_ab.py main=inmerge mergek=mergek mergek=prologue --opt prologue:fuse_qnorm=1 --attn --rounds 3 --workload 384-grid-3x4 > gpurun_out/r05b_ab_cfg5.log 2>&1
python tools/step_ab.py main=inmerge mergek=mergek mergek=prologue --opt prologue:fuse_qnorm=1 --rounds 3 --workload 384-grid-1x2 > gpurun_out/r05b_ab_cfg1.log 2>&1
python bench.py > gpurun_out/r05b_bench.json 2> gpurun_out/r05b_bench.err
tail -n 4 gpurun_out/r05b_pytest.log
cat gpurun_out/r05b_rc.txt gpurun_out/r05b_ab_cfg2.log gpurun_out/r05b_ab_cfg3.log gpurun_out/r05b_ab_cfg5.log gpurun_out/r05b_ab_cfg1.log
tail -c 600 gpurun_out/r05b_bench.json
