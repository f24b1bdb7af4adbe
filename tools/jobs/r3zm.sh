# round 3, call ZM: short-key cross-attention dealt out per wave (five-wave workgroups: 256 workgroups at 1024 queries): tests, A/B vs the previous commit
mkdir -p gpurun_out/r3zm
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_attention_golden_gpu.py -m gpu -q -x -k "attn or attention" 2>&1 | tail -2
for i in 1 2; do
for v in head new; do
  if [ $v = new ]; then unset TMIX_LIB; else export TMIX_LIB=tools/ab/$v/libtmix_hip.so; fi
  TMIX_BENCH_SHAPES=1 timeout 400 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video 2>gpurun_out/r3zm/shapes_$v.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'])"
done; done
grep "'attn'" gpurun_out/r3zm/shapes_head.err gpurun_out/r3zm/shapes_new.err
