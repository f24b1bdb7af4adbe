mkdir -p gpurun_out
timeout 2400 python tools/make_tune_table.py gpurun_out/tuned_r2e.json > gpurun_out/tune_r2e.log 2>&1
echo "tune rc=$?"; tail -3 gpurun_out/tune_r2e.log
cp gpurun_out/tuned_r2e.json tweediemix_amd/tuned_gfx950.json
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_full.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_full.log
bash tools/collect_profile.sh r2a > gpurun_out/collect_r2a.log 2>&1; tail -5 gpurun_out/collect_r2a.log
timeout 1200 python bench.py > gpurun_out/bench_r2e.json 2> gpurun_out/bench_r2e.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r2e.json')); r=d['roofline']
print(round(d['value'],2), round(d['ms_per_step'],2), d['images_per_s'], round(r['achieved']), {k:round(v['sum_launch_ms'],2) for k,v in r['classes'].items()})
for k,v in d['other_configs'].items(): print(k, v['value'], v['parity_check']['rel_l2'])
PY
