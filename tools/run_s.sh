for i in 1 2; do
python bench.py --kind lora --dtype fp8 --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>gpurun_out/s0.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chain', round(d['value'],2), round(d['ms_per_step'],2), d.get('parity_check',{}).get('rel_l2'))"
TMIX_FP8_FF_ROWS=1 python bench.py --kind lora --dtype fp8 --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>gpurun_out/s1.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rows ', round(d['value'],2), round(d['ms_per_step'],2), d.get('parity_check',{}).get('rel_l2'))"
done
python bench.py --kind lora --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>gpurun_out/s2.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 ', round(d['value'],2), round(d['ms_per_step'],2))"
