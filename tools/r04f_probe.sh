export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_text_gpu.py tests/test_model_gpu.py tests/test_vae_gpu.py -q -x -k "gemm or text or t5 or clip or chain or grid or vae" > $OUT/r04f_pytest.log 2>&1; tail -3 $OUT/r04f_pytest.log
python tools/text_bench.py > $OUT/r04f_text_bench.json 2> $OUT/r04f_text_bench.err; tail -c 600 $OUT/r04f_text_bench.json
python tools/vae_bench.py > $OUT/r04f_vae_bench.json 2> $OUT/r04f_vae_bench.err; tail -c 800 $OUT/r04f_vae_bench.json
python tools/e2e_demo.py > $OUT/r04f_e2e.json 2> $OUT/r04f_e2e.err; tail -c 900 $OUT/r04f_e2e.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline > $OUT/r04f_torchrun_bench.json 2> $OUT/r04f_torchrun_bench.err; tail -c 400 $OUT/r04f_torchrun_bench.json
