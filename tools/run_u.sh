python tools/make_tune_table.py gpurun_out/tuned_new.json --refine > gpurun_out/tune.log 2>&1
tail -3 gpurun_out/tune.log
python tools/extend_table_seeds8.py gpurun_out/tuned_new8.json > gpurun_out/tune8.log 2>&1 || true
for i in 1 2 3; do
for t in tweediemix_amd/tuned_gfx950.json gpurun_out/tuned_new.json; do
TMIX_TUNE_FILE=$t python bench.py --kind lora --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', round(d['value'],2), round(d['ms_per_step'],2))"
done; done
