# round 3, call ZR: e4m3 copy inside the straight-line epilogue (fp8 plans' bf16 GEMMs), quad maximum on DPP: fp8 tests, fp8 step A/B vs the previous commit
mkdir -p gpurun_out/r3zr
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "fp8 or e4m3" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -k "fp8" 2>&1 | tail -2
for i in 1 2; do
for v in head new; do
  if [ $v = new ]; then unset TMIX_LIB; else export TMIX_LIB=tools/ab/$v/libtmix_hip.so; fi
  timeout 400 python bench.py --dtype fp8 --kind lora --no-cpu-baseline --no-trajectory --no-video 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'])"
done; done
