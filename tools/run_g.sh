mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for cfg in "1 1" "2 1" "2 2" "4 1" "4 2"; do set -- $cfg
  timeout 900 python bench.py --kind lora --no-trajectory --no-cpu-baseline --seeds-per-gpu $1 --streams $2 > gpurun_out/bench_g_$1_$2.json 2>/dev/null
  python - $1 $2 <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/bench_g_{sys.argv[1]}_{sys.argv[2]}.json')); r=d['roofline']
print('seeds',sys.argv[1],'streams',sys.argv[2], round(d['value'],2), 'seed-steps/s', round(d['ms_per_step'],2), 'ms per seed-step; gemm TF', round(r['achieved']))
PY
done
