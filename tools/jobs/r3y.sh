# round 3, call Y: next-launch weight prefetch (tmix_gemm_prefetch_next) on/off, same box; unet parity tests with it on
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_ops_gpu.py -m gpu -q -x -k "not fp8_projections_full and not headline" 2>&1 | tail -2
for i in 1 2; do
for v in off on; do
  if [ $v = on ]; then unset TMIX_NO_PREFETCH; else export TMIX_NO_PREFETCH=1; fi
  TMIX_BENCH_SHAPES=1 timeout 400 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video 2>gpurun_out/shapes_pf_$v.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prefetch $v', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'])"
done; done
