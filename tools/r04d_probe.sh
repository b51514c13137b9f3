export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
python tools/step_ab.py main nogelu --rounds 5 > $OUT/r04d_gelu_ab.log 2>&1; tail -2 $OUT/r04d_gelu_ab.log
python tools/step_ab.py main=on main=off --opt off:splitk=0 --workload 1024-sdedit-upsample --steps 27 --rounds 5 > $OUT/r04d_splitk_ab_sdedit.log 2>&1; tail -2 $OUT/r04d_splitk_ab_sdedit.log
python tools/step_ab.py main=on main=off --opt off:splitk=0 --workload 1024-sdedit-upsample --steps 27 --rounds 5 --per-gpu-batch 2 > $OUT/r04d_splitk_ab_sdedit_pb2.log 2>&1; tail -2 $OUT/r04d_splitk_ab_sdedit_pb2.log
python tools/step_ab.py main=on main=off --opt off:splitk=0 --workload 384-grid-1x2 --steps 30 --rounds 5 > $OUT/r04d_splitk_ab_cfg1.log 2>&1; tail -2 $OUT/r04d_splitk_ab_cfg1.log
bash tools/measure_round.sh r04d_sdedit 1024-sdedit-upsample > $OUT/r04d_sdedit_measure.log 2>&1; head -8 $OUT/r04d_sdedit_step_breakdown.csv
bash tools/measure_round.sh r04d_cfg1 384-grid-1x2 > $OUT/r04d_cfg1_measure.log 2>&1; head -10 $OUT/r04d_cfg1_step_breakdown.csv
python bench.py --steps 20 --warmup 5 > $OUT/r04d_bench_driver_args.json 2> $OUT/r04d_bench_driver_args.err; tail -c 1500 $OUT/r04d_bench_driver_args.json
