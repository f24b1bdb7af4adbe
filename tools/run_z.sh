timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "attention or attn" 2>&1 | tail -3
timeout 400 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -k "fp8" 2>&1 | tail -3
for i in 1 2; do
timeout 200 python bench.py --kind lora --dtype fp8 --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>gpurun_out/z0.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp8 chain', round(d['value'],2), round(d['ms_per_step'],2), d['parity_check']['rel_l2'])"
TMIX_FP8_FF_ROWS=1 timeout 200 python bench.py --kind lora --dtype fp8 --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>gpurun_out/z1.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp8 rows ', round(d['value'],2), round(d['ms_per_step'],2), d['parity_check']['rel_l2'])"
done
timeout 200 python bench.py --kind lora --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16', round(d['value'],2), round(d['ms_per_step'],2))"
