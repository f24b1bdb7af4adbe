# round 3, call ZY: GroupNorm statistics from the producers in the UNet plan: plan tests, then same-box A/B of the step
#   nocs = library without the column-statistics flavours (their cost to the launches that do not use them), old = this library with the statistics kernel, new = default
mkdir -p gpurun_out/r3zy
timeout 900 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -s -k "statistics_from_the_producers or unet_plan_matches_oracle" 2>&1 | grep -v amdgpu.ids | grep -E "norms from|passed|failed|Error|error" | tail -12
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "producer_partials" 2>&1 | tail -2
one() {  # name, env...
  n=$1; shift
  env "$@" TMIX_BENCH_SHAPES=1 timeout 600 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video --steps 40 2>gpurun_out/r3zy/$n.err | tail -1 > gpurun_out/r3zy/$n.json
  python -c "import sys,json; d=json.loads(open('gpurun_out/r3zy/$n.json').read()); print('$n', round(d['value'],2), round(d['ms_per_step'],3), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'], d['config']['tilings']['follow_shipped_table'])"
}
for r in 1 2; do
  one nocs$r TMIX_LIB=tools/ab/nocs/libtmix_hip.so TMIX_GN_STATS_KERNEL=1
  one old$r TMIX_GN_STATS_KERNEL=1
  one new$r TMIX_X=0
done
grep -A4 "boundaries" gpurun_out/r3zy/new2.err | cut -c1-120
timeout 1200 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -s -k "headline_size_timed_plan and lora-128-1" 2>&1 | grep -E "SDXL|passed|failed|rror" | tail -4
