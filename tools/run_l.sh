mkdir -p gpurun_out
bash tools/collect_profile.sh r2b > gpurun_out/collect_r2b.log 2>&1; tail -3 gpurun_out/collect_r2b.log
timeout 1500 python bench.py > gpurun_out/bench_r2l.json 2> gpurun_out/bench_r2l.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r2l.json')); r=d['roofline']
print(round(d['value'],2), round(d['ms_per_step'],2), d['images_per_s'], round(r['achieved']), r['frac'], r['traffic'])
for k,v in d['other_configs'].items(): print(k, v['value'])
PY
grep -c attn_small gpurun_out/prof_r2b/r2b_kernel_stats.csv
