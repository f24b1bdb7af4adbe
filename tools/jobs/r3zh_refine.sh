# round 3, call ZH: the shipped table refined by graph timing with the new epilogues and tiling 21 among the candidates (single seed, then 4 co-batched), A/B
mkdir -p gpurun_out/r3zh
timeout 2400 python tools/refine_table.py tweediemix_amd/tuned_gfx950.json gpurun_out/r3zh/refined.json --cobatch 4 --top 28 --reps 7 --cands 2,4,5,7,12,14,16,17,18,20,21 > gpurun_out/r3zh/refine.log 2>&1; echo "refine rc=$?"; grep "refined" gpurun_out/r3zh/refine.log
for i in 1 2; do
for tb in tweediemix_amd/tuned_gfx950.json gpurun_out/r3zh/refined.json; do
  TMIX_TUNE_FILE=$tb timeout 400 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tb', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()})"
done; done
