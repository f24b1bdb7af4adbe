# round 3, call ZZD: conv_shortcut inside conv2's launch (shortcut taps), no concat launches -- kernel and plan tests, then same-box A/B (old = shortcut GEMM + concat)
mkdir -p gpurun_out/r3zzd
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "shortcut or conv3x3 or column_statistics or reject" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -s -k "shortcut_in_the_conv or statistics_from_the_producers or unet_plan_matches_oracle or row_split" 2>&1 | grep -v amdgpu.ids | grep -E "ops;|norms from|passed|failed|Error|error" | tail -12
one() {  # name, env...
  n=$1; shift
  env "$@" TMIX_BENCH_SHAPES=1 timeout 600 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video --steps 40 2>gpurun_out/r3zzd/$n.err | tail -1 > gpurun_out/r3zzd/$n.json
  python -c "import sys,json; d=json.loads(open('gpurun_out/r3zzd/$n.json').read()); print('$n', round(d['value'],2), round(d['ms_per_step'],3), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'], d['config']['tilings']['follow_shipped_table'], round(d['roofline']['frac'],3))"
}
for r in 1 2; do
  one old$r TMIX_SHORTCUT_GEMM=1
  one new$r TMIX_X=0
done
grep -A9 "boundaries" gpurun_out/r3zzd/new2.err | cut -c1-120
grep "'conv'" gpurun_out/r3zzd/new2.err | cut -c1-120
timeout 1200 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -s -k "headline_size_timed_plan and (lora-128-1 or custom-128-2)" 2>&1 | grep -E "SDXL|passed|failed|rror" | tail -4
