#!/bin/bash
# round 5, GPU call 11: the driver's form of the bench line, cfg 1 through the product sampler, end-to-end demo
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r05y_bench_driver_args.json 2> gpurun_out/r05y_bench_driver_args.err
python tools/sampler_rate.py --workload 384-grid-1x2 --samples 60 > gpurun_out/r05y_sampler_rate_cfg1.json 2>&1
python tools/sampler_rate.py --workload 384-grid-2x3 --samples 8 > gpurun_out/r05y_sampler_rate_cfg2.json 2>&1
python tools/e2e_demo.py > gpurun_out/r05y_e2e.json 2> gpurun_out/r05y_e2e.err
tail -c 400 gpurun_out/r05y_bench_driver_args.json; grep -hv amdgpu.ids gpurun_out/r05y_sampler_rate_cfg1.json gpurun_out/r05y_sampler_rate_cfg2.json; tail -c 600 gpurun_out/r05y_e2e.json
