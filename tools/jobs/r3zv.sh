# round 3, call ZV: LoRA low-rank form at the SDXL widths: oracle test, then the fusion step merged vs low-rank on this box
mkdir -p gpurun_out/r3zv
timeout 1500 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -s -k "low_rank_lora_full_size" 2>&1 | tail -4
cp tweediemix_amd/tuned_gfx950.json gpurun_out/r3zv/table_lr.json
for v in merged lowrank; do
  TMIX_TUNE_FILE=gpurun_out/r3zv/table_lr.json TMIX_BENCH_SHAPES=1 timeout 900 python bench.py --kind lora --lora-mode $v --no-cpu-baseline --no-trajectory --no-video 2>gpurun_out/r3zv/shapes_$v.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'], d['roofline']['launches_per_step'])"
done
nvidia-smi >/dev/null 2>&1; python - <<'PY'
import torch; print('peak mem MB', torch.cuda.max_memory_allocated()/1e6)
PY
