mkdir -p gpurun_out
timeout 2400 python tools/make_tune_table.py gpurun_out/tuned_r2k.json > gpurun_out/tune_r2k.log 2>&1
echo "tune rc=$?"; tail -2 gpurun_out/tune_r2k.log
for t in tweediemix_amd/tuned_gfx950.json gpurun_out/tuned_r2k.json; do
  TMIX_TUNE_FILE=$t timeout 900 python bench.py --kind lora --no-trajectory --no-cpu-baseline > gpurun_out/bench_k_$(basename $t .json).json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_k_*.json')):
    d=json.load(open(f)); r=d['roofline']; print(f, round(d['value'],2), round(d['ms_per_step'],2), round(r['achieved']), {k:round(v['sum_launch_ms'],2) for k,v in r['classes'].items()})
import collections
t=json.load(open('gpurun_out/tuned_r2k.json')); print(sorted(collections.Counter(t.values()).items()))
PY
