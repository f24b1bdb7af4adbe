# round 3, call ZC: hardware bf16 conversion, whole-step A/B on one box (swcvt = the integer form), then the kernel tests
mkdir -p gpurun_out/r3zc
for i in 1 2; do
for v in swcvt new; do
  if [ $v = new ]; then unset TMIX_LIB; else export TMIX_LIB=tools/ab/$v/libtmix_hip.so; fi
  TMIX_BENCH_SHAPES=1 timeout 400 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video 2>gpurun_out/r3zc/shapes_$v.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'])"
done; done
unset TMIX_LIB
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_attention_golden_gpu.py tests/test_vae_gpu.py -m gpu -q -x 2>&1 | tail -3
