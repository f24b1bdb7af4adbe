# round 3, call ZZC: the whole -m gpu suite + smoke + the driver's default bench command on the current tree (producer-side GroupNorm statistics, epilogue-family kernels)
mkdir -p gpurun_out/r3zzc
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r3zzc/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3zzc/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke" | tail -2
timeout 1500 python bench.py > gpurun_out/r3zzc/bench.json 2> gpurun_out/r3zzc/bench.err; echo "bench rc=$? lines=$(wc -l < gpurun_out/r3zzc/bench.json)"
python - <<'PY'
import json; d=json.loads(open('gpurun_out/r3zzc/bench.json').read())
print('value', round(d['value'],2), 'ms', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],3), 'img/s', round(d['images_per_s'],3), 'traj', round(d['trajectory_steps_per_s'],2), 'vae', round(d['vae_decode_ms'],2))
print({k:(round(v['value'],2), round(v['ms_per_step'],2)) for k,v in d['other_configs'].items()})
print({k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['config']['tilings']['follow_shipped_table'], d['dist']['ranks_seen'])
print(d['trajectory']['single_image'])
PY
