# round 3, call I: the new parity tests (timed plans vs oracle, fp8 at 128^2, full-size VAE, one-rank RCCL), the smoke, the default bench line
mkdir -p gpurun_out/r3i
timeout 2400 python -m pytest tests/test_unet_gpu.py tests/test_vae_gpu.py tests/test_sampler_gpu.py -m gpu -q -x -s -k "headline or fp8 or full_size_vae or rccl" 2>&1 | grep -E "rel_l2|max abs|passed|failed|Error|error|assert" | head -40
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 1500 python bench.py > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r3i/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3i/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('images_per_s'), d.get('vae_decode_ms'), d.get('trajectory_steps_per_s'))
print(d['dist']); print(d['config']['tilings']); print({k:(v.get('value'),v.get('ms_per_step')) for k,v in d['other_configs'].items()}); print(d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:200])
PY
