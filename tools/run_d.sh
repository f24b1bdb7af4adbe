mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py tests/test_attention_golden_gpu.py tests/test_sampler_gpu.py -m gpu -q -s 2>&1 | grep -E "passed|failed|FAILED|rel_l2|rel-L2|Error" > gpurun_out/pytest_d.log
tail -20 gpurun_out/pytest_d.log
TMIX_BENCH_SHAPES=1 timeout 900 python bench.py --kind custom --no-trajectory --no-cpu-baseline --streams 1 > gpurun_out/bench_d1.json 2> gpurun_out/bench_d1.err
TMIX_ATTN_GENERAL=1 timeout 900 python bench.py --kind custom --no-trajectory --no-cpu-baseline --streams 1 > gpurun_out/bench_d1g.json 2> gpurun_out/bench_d1g.err
timeout 900 python bench.py --kind custom --no-trajectory --no-cpu-baseline --streams 2 > gpurun_out/bench_d2.json 2> gpurun_out/bench_d2.err
timeout 900 python bench.py --kind custom --no-trajectory --no-cpu-baseline --streams 1 --dtype fp8 > gpurun_out/bench_d1f.json 2> gpurun_out/bench_d1f.err
for f in d1 d1g d2 d1f; do python - $f <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/bench_{sys.argv[1]}.json')); r=d['roofline']
print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],2), 'gemm TF', round(r['achieved']), {k:round(v['sum_launch_ms'],2) for k,v in r['classes'].items()}, d['parity_check']['rel_l2'])
PY
done
grep "attn" gpurun_out/bench_d1.err | head -6
