for i in 1 2; do
python bench.py --kind lora --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', round(d['value'],2), round(d['ms_per_step'],2))"
HIP_FORCE_DEV_KERNARG=1 python bench.py --kind lora --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('devkernarg', round(d['value'],2), round(d['ms_per_step'],2))"
done
python bench.py --kind custom --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('custom', round(d['value'],2), round(d['ms_per_step'],2))"
python bench.py --kind lora --dtype fp8 --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp8', round(d['value'],2), round(d['ms_per_step'],2))"
