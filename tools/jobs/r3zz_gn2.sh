# round 3, call ZZ: partial-combining kernel on 16 waves; VAE plans on the producer statistics; same-box A/B (old = statistics kernel)
mkdir -p gpurun_out/r3zz
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "producer_partials or through_the_partials" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_vae_gpu.py -m gpu -q -x 2>&1 | tail -3
one() {  # name, env...
  n=$1; shift
  env "$@" TMIX_BENCH_SHAPES=1 timeout 600 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video --steps 40 2>gpurun_out/r3zz/$n.err | tail -1 > gpurun_out/r3zz/$n.json
  python -c "import sys,json; d=json.loads(open('gpurun_out/r3zz/$n.json').read()); print('$n', round(d['value'],2), round(d['ms_per_step'],3), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'], d['config']['tilings']['follow_shipped_table'])"
}
for r in 1 2; do
  one old$r TMIX_GN_STATS_KERNEL=1
  one new$r TMIX_X=0
done
grep "norm'" gpurun_out/r3zz/new2.err | cut -c1-110
for v in 1 0; do echo "vae TMIX_GN_STATS_KERNEL=$v"; if [ $v = 1 ]; then TMIX_GN_STATS_KERNEL=1 timeout 600 python tools/vae_time.py 2>&1 | tail -3; else timeout 600 python tools/vae_time.py 2>&1 | tail -3; fi; done
