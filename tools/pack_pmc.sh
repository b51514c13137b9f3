export TMPDIR=/tmp
OUT=gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr -d $OUT/r04i_packpmc/$ctr -o p --output-format csv -- python tools/pack_bench.py --only streaming --iters 20 > $OUT/r04i_packpmc_$ctr.log 2>&1
done
python tools/pmc_summary.py $OUT/r04i_packpmc/*/p_counter_collection.csv > $OUT/r04i_pack_pmc_summary.json
rm -rf $OUT/r04i_packpmc
python - <<PY
import json
d = json.load(open("$OUT/r04i_pack_pmc_summary.json"))
for k, v in d.items():
    f, w, t = 2 * v.get("FETCH_SIZE", 0) * 1024, v.get("WRITE_SIZE", 0) * 1024, v["_dur_ns"]
    print(f"{k:24s} fetch {f/1e6:8.1f} MB (x2 gfx950 correction)  write {w/1e6:8.1f} MB  {t/1e3:7.1f} us  -> {(f+w)/t:7.1f} GB/s  ({v['dispatches']} dispatches)")
PY
