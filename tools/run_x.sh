timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm" 2>&1 | tail -3
timeout 400 python -m pytest tests/test_unet_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 200 python bench.py --kind lora --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lora', round(d['value'],2), round(d['ms_per_step'],2), d['parity_check']['rel_l2'])"
