# round 3, call W: write-through (sc1) output stores in the GEMM / conv epilogues vs plain stores: kernel boundaries and the step
for i in 1 2; do
for v in base wt; do
  if [ $v = base ]; then unset TMIX_LIB; else export TMIX_LIB=tools/ab/$v/libtmix_hip.so; fi
  TMIX_BENCH_SHAPES=1 timeout 400 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video 2>gpurun_out/shapes_$v.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'])"
  grep "boundaries" gpurun_out/shapes_$v.err; grep "gemm  -> gemm" gpurun_out/shapes_$v.err
done; done
