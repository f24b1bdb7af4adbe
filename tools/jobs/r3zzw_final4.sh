# round 3, call ZZW: last verification of the round (short-key attention prefetch on top of ZZT): the whole -m gpu suite, smoke, the default bench, profile set r3f, tmix vs hipBLASLt on this path's shapes
mkdir -p gpurun_out/r3zzw
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r3zzw/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3zzw/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke" | tail -2
timeout 1500 python bench.py > gpurun_out/r3zzw/bench.json 2> gpurun_out/r3zzw/bench.err; echo "bench rc=$? lines=$(wc -l < gpurun_out/r3zzw/bench.json)"
python - <<'PY'
import json; d=json.loads(open('gpurun_out/r3zzw/bench.json').read())
print('value', round(d['value'],2), 'ms', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],3), 'achieved', round(d['roofline']['achieved'],1), 'img/s', round(d['images_per_s'],3), 'traj', round(d['trajectory_steps_per_s'],2), 'vae', round(d['vae_decode_ms'],2))
print({k:(round(v['value'],2), round(v['ms_per_step'],2)) for k,v in d['other_configs'].items()})
print({k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['config']['tilings']['follow_shipped_table'], d['dist']['ranks_seen'])
print(d['trajectory']['single_image'])
PY
bash tools/collect_profile.sh r3f > gpurun_out/r3zzw/collect.log 2>&1; tail -3 gpurun_out/r3zzw/collect.log; rm -f gpurun_out/prof_r3f/*.log
timeout 600 python tools/vs_blas.py gpurun_out/r3zzw/vs_blas_hot.json 2>&1 | grep -v amdgpu.ids | tail -10
