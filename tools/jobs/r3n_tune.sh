# round 3, call N: the shipped tile table re-measured from scratch with this round's kernels (in-situ timing, every sampler plan)
mkdir -p gpurun_out/r3n
TMIX_TUNE_REPS=3 timeout 3300 python tools/make_tune_table.py gpurun_out/r3n/tuned_r3.json > gpurun_out/r3n/tune.log 2>&1
echo "tune rc=$?"; tail -4 gpurun_out/r3n/tune.log
TMIX_TUNE_FILE=gpurun_out/r3n/tuned_r3.json timeout 600 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new table', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()})"
