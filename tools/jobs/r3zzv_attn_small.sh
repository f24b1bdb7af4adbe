# round 3, call ZZV: short-key attention with both query blocks requested up front (one memory round trip instead of two): tests, same-box A/B
mkdir -p gpurun_out/r3zzv
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_attention_golden_gpu.py -m gpu -q -x -k "attention or attn or hook or golden" 2>&1 | tail -2
one() {  # name, env...
  n=$1; shift
  env "$@" TMIX_BENCH_SHAPES=1 timeout 600 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video --steps 40 2>gpurun_out/r3zzv/$n.err | tail -1 > gpurun_out/r3zzv/$n.json
  python -c "import sys,json; d=json.loads(open('gpurun_out/r3zzv/$n.json').read()); print('$n', round(d['value'],2), round(d['ms_per_step'],3), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()})"
  grep -E "'attn', 4, 20, 1024, 77|'attn', 4, 10, 4096, 77" gpurun_out/r3zzv/$n.err | cut -c1-110
}
for r in 1 2; do
  one head$r TMIX_LIB=tools/ab/head/libtmix_hip.so
  one new$r TMIX_X=0
done
