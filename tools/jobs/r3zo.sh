# round 3, call ZO: sched_barrier between epilogue chunks (256x320: 75 -> 22 spills), A/B vs the previous commit
mkdir -p gpurun_out/r3zo
for i in 1 2; do
for v in head new; do
  if [ $v = new ]; then unset TMIX_LIB; else export TMIX_LIB=tools/ab/$v/libtmix_hip.so; fi
  TMIX_BENCH_SHAPES=1 timeout 400 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video 2>gpurun_out/r3zo/shapes_$v.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'])"
done; done
grep "10240, 1280\|5120, 640" gpurun_out/r3zo/shapes_head.err gpurun_out/r3zo/shapes_new.err
