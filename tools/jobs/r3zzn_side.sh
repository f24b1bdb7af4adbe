# round 3, call ZZN: the time-embedding launches on a side stream beside conv_in / the first norm (fork / join inside the captured step): tests, then A/B
mkdir -p gpurun_out/r3zzn
timeout 900 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -k "unet_plan_matches_oracle or row_split" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_sampler_gpu.py -m gpu -q -x 2>&1 | tail -2
for r in 1 2 3; do
for v in 1 0; do
  if [ $v = 1 ]; then export TMIX_NO_SIDE_STREAM=1; else unset TMIX_NO_SIDE_STREAM; fi
  timeout 600 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no_side=$v', round(d['value'],2), round(d['ms_per_step'],3), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'])"
done; done
