# round 3, call ZZB: single-flavour instantiations (EK 4, 5, 6) against the epilogue-family commit, same box
mkdir -p gpurun_out/r3zzb
timeout 1500 python -m pytest tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -3
one() {  # name, env...
  n=$1; shift
  env "$@" TMIX_BENCH_SHAPES=1 timeout 600 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video --steps 40 2>gpurun_out/r3zzb/$n.err | tail -1 > gpurun_out/r3zzb/$n.json
  python -c "import sys,json; d=json.loads(open('gpurun_out/r3zzb/$n.json').read()); print('$n', round(d['value'],2), round(d['ms_per_step'],3), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'], d['config']['tilings']['follow_shipped_table'])"
}
for r in 1 2; do
  one head$r TMIX_LIB=tools/ab/head/libtmix_hip.so
  one new$r TMIX_X=0
done
for n in head2 new2; do echo $n; grep -E "'gemm', 1, 4096, 10240|'gemm', 4, 1024, 1280, 1280|'gemm', 1, 4096, 1280, 5120|'gemm', 4, 1024, 3840|'conv', 4, 32, 32, 1280, 1280, 0|'conv', 4, 128, 128, 320, 320, 0|'conv', 4, 64, 64, 640, 640, 2" gpurun_out/r3zzb/$n.err | cut -c1-120; done
