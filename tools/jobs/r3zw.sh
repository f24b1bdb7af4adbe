# round 3, call ZW: tmix_lora_down on MFMA (16 rows per workgroup, K split over four waves): tests, then merged vs low-rank step
mkdir -p gpurun_out/r3zw
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "lora_down" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -s -k "low_rank" 2>&1 | grep -v amdgpu.ids | tail -8
cp tweediemix_amd/tuned_gfx950.json gpurun_out/r3zw/table_lr.json
for v in merged lowrank; do
  TMIX_TUNE_FILE=gpurun_out/r3zw/table_lr.json TMIX_BENCH_SHAPES=1 timeout 900 python bench.py --kind lora --lora-mode $v --no-cpu-baseline --no-trajectory --no-video 2>gpurun_out/r3zw/shapes_$v.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'])"
done
grep -A3 "boundaries" gpurun_out/r3zw/shapes_lowrank.err | cut -c1-110
