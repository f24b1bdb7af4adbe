# round 3, call ZK: residual rows requested two K-tiles before the end: A/B vs the previous commit, lab check
mkdir -p gpurun_out/r3zk
L=tools/gemm_lab/lab
timeout 300 $L check nocold 1024,1280,1280,1,br 2048,2560,1280,1,brs 520,648,64,1,brs 520,640,64,1,br 520,640,128,1,br 520,640,192,1,br cfgs=2,3,5,12,13,15,18,19,20,21 reps=3 2>&1 | grep -c " ok"
for i in 1 2; do
for v in head new; do
  if [ $v = new ]; then unset TMIX_LIB; else export TMIX_LIB=tools/ab/$v/libtmix_hip.so; fi
  TMIX_BENCH_SHAPES=1 timeout 400 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video 2>gpurun_out/r3zk/shapes_$v.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'])"
done; done
