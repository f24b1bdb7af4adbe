# round 3, call ZZK: final state -- the whole -m gpu suite, smoke, the driver's default bench command, old / new tile table A/B, profile set r3d
mkdir -p gpurun_out/r3zzk
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r3zzk/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3zzk/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke" | tail -2
timeout 1500 python bench.py > gpurun_out/r3zzk/bench.json 2> gpurun_out/r3zzk/bench.err; echo "bench rc=$? lines=$(wc -l < gpurun_out/r3zzk/bench.json)"
python - <<'PY'
import json; d=json.loads(open('gpurun_out/r3zzk/bench.json').read())
print('value', round(d['value'],2), 'ms', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],3), 'achieved', round(d['roofline']['achieved'],1), 'img/s', round(d['images_per_s'],3), 'traj', round(d['trajectory_steps_per_s'],2), 'vae', round(d['vae_decode_ms'],2))
print({k:(round(v['value'],2), round(v['ms_per_step'],2)) for k,v in d['other_configs'].items()})
print({k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['config']['tilings']['follow_shipped_table'], d['dist']['ranks_seen'])
print(d['trajectory']['single_image'])
PY
for i in 1 2; do
for tb in tools/ab/old_table.json tweediemix_amd/tuned_gfx950.json; do
  TMIX_TUNE_FILE=$tb TMIX_BENCH_SHAPES=1 timeout 600 python bench.py --kind lora --no-cpu-baseline --no-video --no-trajectory --steps 40 2>gpurun_out/r3zzk/shapes_$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tb', round(d['value'],2), round(d['ms_per_step'],3), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()})"
done; done
grep -B1 -A10 "boundaries" gpurun_out/r3zzk/shapes_2.err | cut -c1-120
bash tools/collect_profile.sh r3d > gpurun_out/r3zzk/collect.log 2>&1; tail -3 gpurun_out/r3zzk/collect.log; rm -f gpurun_out/prof_r3d/*.log
