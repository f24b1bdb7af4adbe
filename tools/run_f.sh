mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_sampler_gpu.py -m gpu -q -k "bench or cli" > gpurun_out/pytest_f.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_f.log
( time timeout 1500 python bench.py > gpurun_out/bench_r2f.json 2> gpurun_out/bench_r2f.err ) 2>&1 | tail -3; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r2f.json')); r=d['roofline']
print(round(d['value'],2), round(d['ms_per_step'],2), d['images_per_s'], round(r['achieved']), r['frac'])
print(d['trajectory'])
for k,v in d['other_configs'].items(): print(k, v['value'], v['parity_check']['rel_l2'])
print(d['cpu_baseline']['value'], d['cpu_baseline']['config1'])
PY
