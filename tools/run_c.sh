mkdir -p gpurun_out
L=tools/gemm_lab/lab
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py tests/test_attention_golden_gpu.py -m gpu -q > gpurun_out/pytest_c.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/pytest_c.log; grep -E "rel-L2|rel_l2" gpurun_out/pytest_c.log | head
timeout 600 $L 1024,3840,1280,4,bt 1024,3840,1280,4,btf 4096,1280,1280,1,br 4096,1280,1280,1,brf 16384,5120,640,1,g 16384,5120,640,1,gf 8192,8192,8192 8192,8192,8192,1,f cfgs=2,14,16,17 reps=20 > gpurun_out/lab7.txt 2>&1
grep -v timeline gpurun_out/lab7.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_r2c.json 2> gpurun_out/bench_r2c.err; echo "bench rc=$?"; tail -3 gpurun_out/bench_r2c.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r2c.json'))
print({k:d[k] for k in ('value','ms_per_step','images_per_s')})
for k,v in d['other_configs'].items(): print(k, v['value'], v['parity_check']['rel_l2'])
r=d['roofline']; print(r['achieved'], r['graph_replay_ms'], r['uninstrumented_graph_replay_ms'])
PY
