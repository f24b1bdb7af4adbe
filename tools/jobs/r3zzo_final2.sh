# round 3, call ZZO: final state again (kernel-argument touch, SiLU(emb) stored once): the whole -m gpu suite, smoke, the default bench
mkdir -p gpurun_out/r3zzo
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r3zzo/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3zzo/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke" | tail -2
timeout 1500 python bench.py > gpurun_out/r3zzo/bench.json 2> gpurun_out/r3zzo/bench.err; echo "bench rc=$? lines=$(wc -l < gpurun_out/r3zzo/bench.json)"
python - <<'PY'
import json; d=json.loads(open('gpurun_out/r3zzo/bench.json').read())
print('value', round(d['value'],2), 'ms', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],3), 'achieved', round(d['roofline']['achieved'],1), 'img/s', round(d['images_per_s'],3), 'traj', round(d['trajectory_steps_per_s'],2), 'vae', round(d['vae_decode_ms'],2))
print({k:(round(v['value'],2), round(v['ms_per_step'],2)) for k,v in d['other_configs'].items()})
print({k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['config']['tilings']['follow_shipped_table'], d['dist']['ranks_seen'])
PY
for v in 1 0 1 0; do
  if [ $v = 1 ]; then git_stash=1; fi
  timeout 600 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run', round(d['value'],2), round(d['ms_per_step'],3))"
done
