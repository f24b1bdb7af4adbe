# round 3, call ZZU: tiling 20 offered to the convolutions of the co-batched (4 seeds per launch) plans, then images/s old / new table
mkdir -p gpurun_out/r3zzu
timeout 1500 python tools/refine_table.py tweediemix_amd/tuned_gfx950.json gpurun_out/r3zzu/refined.json --cobatch 4 --only-cobatch --top 60 --reps 7 --kinds lora --cands 20 > gpurun_out/r3zzu/refine.log 2>&1; echo "refine rc=$?"
grep -E "refine \(" gpurun_out/r3zzu/refine.log | awk -F': ' '{split($2,a," "); if (a[1]!=a[3]) print}' | cut -c1-170; grep refined gpurun_out/r3zzu/refine.log
for i in 1 2; do
for tb in tweediemix_amd/tuned_gfx950.json gpurun_out/r3zzu/refined.json; do
  TMIX_TUNE_FILE=$tb timeout 600 python bench.py --kind lora --no-cpu-baseline --no-video 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tb', round(d['value'],2), round(d['ms_per_step'],2), 'img/s', round(d['images_per_s'],4), d['config']['tilings']['follow_shipped_table'])"
done; done
