for i in 1 2 3; do
python bench.py --kind lora --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', round(d['value'],2), round(d['ms_per_step'],2))"
TMIX_NT_STORES=1 python bench.py --kind lora --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nt  ', round(d['value'],2), round(d['ms_per_step'],2))"
done
TMIX_NT_STORES=1 TMIX_BENCH_SHAPES=1 python bench.py --kind lora --no-trajectory --no-cpu-baseline --steps 10 --warmup 2 2> gpurun_out/shapes_nt.err > /dev/null; grep -A4 "boundaries between" gpurun_out/shapes_nt.err
