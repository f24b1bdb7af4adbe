# round 3, call ZE: straight-line staged epilogue, whole-step A/B on one box (abl128 = generic epilogue only), per-shape in situ
mkdir -p gpurun_out/r3ze
for i in 1 2; do
for v in abl128 new; do
  if [ $v = new ]; then unset TMIX_LIB; else export TMIX_LIB=tools/ab/$v/libtmix_hip.so; fi
  TMIX_BENCH_SHAPES=1 timeout 400 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video 2>gpurun_out/r3ze/shapes_$v.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'])"
done; done
