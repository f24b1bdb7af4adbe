# round 3, call ZZR: the convolution on tiling 20 (128 x 160 + two loader waves): kernel tests, a graph-timed refine that offers it to the conv shapes, A/B
mkdir -p gpurun_out/r3zzr
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "conv3x3 or conv_leaves or conv_with_shortcut or conv_temporal" 2>&1 | tail -3
timeout 1500 python tools/refine_table.py tweediemix_amd/tuned_gfx950.json gpurun_out/r3zzr/refined.json --top 40 --reps 9 --kinds lora --cands 20 > gpurun_out/r3zzr/refine.log 2>&1; echo "refine rc=$?"
grep -E "refine \('conv" gpurun_out/r3zzr/refine.log | awk -F': ' '{split($2,a," "); if (a[1]!=a[3]) print}' | cut -c1-170; grep refined gpurun_out/r3zzr/refine.log
for i in 1 2 3; do
for tb in tweediemix_amd/tuned_gfx950.json gpurun_out/r3zzr/refined.json; do
  TMIX_TUNE_FILE=$tb timeout 600 python bench.py --kind lora --no-cpu-baseline --no-video --no-trajectory --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tb', round(d['value'],2), round(d['ms_per_step'],3), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'])"
done; done
