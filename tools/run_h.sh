mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_attention_golden_gpu.py tests/test_unet_gpu.py -m gpu -q -k "attn or attention or unet or headline" 2>&1 | tail -4
for wv in 4 8; do
  TMIX_ATTN_WAVES=$wv TMIX_BENCH_SHAPES=1 timeout 900 python bench.py --kind lora --no-trajectory --no-cpu-baseline > gpurun_out/bench_h_$wv.json 2> gpurun_out/bench_h_$wv.err
  python - $wv <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/bench_h_{sys.argv[1]}.json')); r=d['roofline']
print('waves',sys.argv[1], round(d['value'],2), round(d['ms_per_step'],2), {k:round(v['sum_launch_ms'],2) for k,v in r['classes'].items()})
PY
  grep "attn" gpurun_out/bench_h_$wv.err | head -4
done
