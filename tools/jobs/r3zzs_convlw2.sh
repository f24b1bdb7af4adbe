# round 3, call ZZS: loader-wave convolution after the row-mapping fix (each loader's r-th instruction covers rows slot_a(r) * 8 ..): kernel tests, plan tests incl. the
# headline-size oracle test on the new table, A/B old / new table
mkdir -p gpurun_out/r3zzs
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "conv" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -s -k "unet_plan_matches_oracle or shortcut_in_the_conv or (headline_size_timed_plan and (lora-128-1 or custom-128-1 or lora-64-1))" 2>&1 | grep -E "SDXL|passed|failed|rror" | tail -6
for i in 1 2; do
for tb in tools/ab/old_table.json tweediemix_amd/tuned_gfx950.json; do
  TMIX_TUNE_FILE=$tb timeout 600 python bench.py --kind lora --no-cpu-baseline --no-video --no-trajectory --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tb', round(d['value'],2), round(d['ms_per_step'],3), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'])"
done; done
