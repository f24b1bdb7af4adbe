# round 3, call ZZL: every kernel-argument line touched at kernel entry (the prologue's ~20 serial s_load round trips become scalar-cache hits) vs not, same box
mkdir -p gpurun_out/r3zzl
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm_plain or conv3x3 or attention" 2>&1 | tail -2
one() {  # name, env...
  n=$1; shift
  env "$@" TMIX_BENCH_SHAPES=1 timeout 600 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video --steps 40 2>gpurun_out/r3zzl/$n.err | tail -1 > gpurun_out/r3zzl/$n.json
  python -c "import sys,json; d=json.loads(open('gpurun_out/r3zzl/$n.json').read()); print('$n', round(d['value'],2), round(d['ms_per_step'],3), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'])"
}
for r in 1 2 3; do
  one old$r TMIX_LIB=tools/ab/noka/libtmix_hip.so
  one new$r TMIX_X=0
done
for n in old3 new3; do echo $n; grep -E "'gemm', 1, 4096, 10240|'gemm', 4, 1024, 1280, 1280|'gemm', 1, 4096, 1280, 5120|'gemm', 4, 1024, 3840|'attn', 4, 20, 1024, 1024|'attn', 4, 20, 1024, 77|'conv', 4, 32, 32, 1280, 1280, 0" gpurun_out/r3zzl/$n.err | cut -c1-120; done
