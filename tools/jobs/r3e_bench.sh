# round 3, call E: unit tests of the GEMM/conv kernels with the pipelined loop, then the step bench A/B (old loop = tools/ab/pipe0)
mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do
for v in pipe0 new; do
  if [ $v = new ]; then unset TMIX_LIB; else export TMIX_LIB=tools/ab/$v/libtmix_hip.so; fi
  TMIX_BENCH_SHAPES=1 timeout 300 python bench.py --kind lora --steps 20 --warmup 3 --no-cpu-baseline --no-trajectory 2>gpurun_out/r3e/shapes_$v.err | tail -1 > gpurun_out/r3e/bench_$v.json
  python -c "import json; d=json.load(open('gpurun_out/r3e/bench_$v.json')); print('$v', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, round(d['roofline']['frac'],3))"
done; done
