# round 3, call ZZG: the co-batched (4 seeds per launch, B = 16) entries of the tile table refined by graph timing with this round's kernels, then images/s old / new table
mkdir -p gpurun_out/r3zzg
timeout 1500 python tools/refine_table.py tweediemix_amd/tuned_gfx950.json gpurun_out/r3zzg/refined.json --cobatch 4 --only-cobatch --top 28 --reps 7 --kinds lora > gpurun_out/r3zzg/refine.log 2>&1; echo "refine rc=$?"; grep -E "refined|->" gpurun_out/r3zzg/refine.log | tail -30
for i in 1 2; do
for tb in tweediemix_amd/tuned_gfx950.json gpurun_out/r3zzg/refined.json; do
  TMIX_TUNE_FILE=$tb timeout 600 python bench.py --kind lora --no-cpu-baseline --no-video 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tb', round(d['value'],2), round(d['ms_per_step'],2), 'img/s', round(d['images_per_s'],4), d['config']['tilings']['follow_shipped_table'])"
done; done
