# round 3, call ZQ: fp8 roofline block sanity, then the profile set r3b of the current library (kernel trace bf16 + fp8, FETCH / WRITE / MfmaUtil passes)
mkdir -p gpurun_out/r3zq
timeout 600 python bench.py --no-cpu-baseline --no-trajectory --no-video > gpurun_out/r3zq/bench.json 2> gpurun_out/r3zq/bench.err; echo "bench rc=$?"
python - <<'PY' || exit 1
import json; d=json.loads(open('gpurun_out/r3zq/bench.json').read())
r=d['other_configs']['fp8']['roofline']; print('fp8 roofline', round(r['achieved'],1), round(r['frac'],3), r['launches_per_step'], round(r['sum_launch_ms'],2), r.get('bf16_gemms_of_this_plan'))
print('bf16', round(d['value'],2), round(d['roofline']['frac'],3), 'fp8', round(d['other_configs']['fp8']['value'],2))
PY
bash tools/collect_profile.sh r3b > gpurun_out/r3zq/collect.log 2>&1; tail -5 gpurun_out/r3zq/collect.log
