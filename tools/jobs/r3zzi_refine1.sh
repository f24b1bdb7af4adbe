# round 3, call ZZI: single-seed entries of the tile table refined by graph timing with this round's kernels (the 256 x 320 convolution no longer spills), A/B
mkdir -p gpurun_out/r3zzi
timeout 1500 python tools/refine_table.py tweediemix_amd/tuned_gfx950.json gpurun_out/r3zzi/refined.json --top 30 --reps 9 --kinds lora --cands 2,4,7,12,13,14,19,20,21 > gpurun_out/r3zzi/refine.log 2>&1; echo "refine rc=$?"; grep -E "refined|->" gpurun_out/r3zzi/refine.log | awk '{ if ($0 ~ /refined/ || $(NF-5) != $(NF-3)) print }' | tail -30
for i in 1 2 3; do
for tb in tweediemix_amd/tuned_gfx950.json gpurun_out/r3zzi/refined.json; do
  TMIX_TUNE_FILE=$tb timeout 600 python bench.py --kind lora --no-cpu-baseline --no-video --no-trajectory --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tb', round(d['value'],2), round(d['ms_per_step'],3), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()})"
done; done
