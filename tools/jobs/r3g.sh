mkdir -p gpurun_out/r3g
timeout 900 python tools/quick_tune.py gpurun_out/r3g/tuned_quick.json lora 2>&1 | tail -3
for i in 1 2; do
  TMIX_LIB=tools/ab/pipe0/libtmix_hip.so timeout 300 python bench.py --kind lora --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipe0/shipped-table', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()})"
  TMIX_TUNE_FILE=gpurun_out/r3g/tuned_quick.json TMIX_BENCH_SHAPES=1 timeout 300 python bench.py --kind lora --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>gpurun_out/r3g/shapes_new.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new/quick-table', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()})"
done
