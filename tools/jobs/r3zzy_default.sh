# round 3, call ZZY: the driver's default bench command with the new defaults (50 steps behind 10 warm-up steps)
mkdir -p gpurun_out/r3zzy
( time timeout 1500 python bench.py > gpurun_out/r3zzy/bench.json 2> gpurun_out/r3zzy/bench.err ) 2>&1 | grep real; echo "bench rc=$? lines=$(wc -l < gpurun_out/r3zzy/bench.json)"
python - <<'PY'
import json; d=json.loads(open('gpurun_out/r3zzy/bench.json').read())
print('value', round(d['value'],2), 'ms', round(d['ms_per_step'],2), 'steps', d['steps'], 'warmup', d['warmup'], 'frac', round(d['roofline']['frac'],3), 'achieved', round(d['roofline']['achieved'],1), 'img/s', round(d['images_per_s'],3), 'traj', round(d['trajectory_steps_per_s'],2), 'vae', round(d['vae_decode_ms'],2))
print({k:(round(v['value'],2), round(v['ms_per_step'],2)) for k,v in d['other_configs'].items()})
print({k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['config']['tilings']['follow_shipped_table'], d['dist']['ranks_seen'], d['cpu_baseline']['value'])
PY
