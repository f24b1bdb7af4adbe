mkdir -p gpurun_out/r3h
for s in 1 2; do
  TMIX_TUNE_FILE=gpurun_out/r3g/tuned_quick.json timeout 400 python bench.py --kind lora --streams $s --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $s quick-table', round(d['value'],2), round(d['ms_per_step'],2))"
  timeout 400 python bench.py --kind lora --streams $s --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $s shipped-table', round(d['value'],2), round(d['ms_per_step'],2))"
done
