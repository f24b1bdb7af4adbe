# round 3, call Q: graph-timed refinement of the re-measured tile table (single-seed plans, then 4 co-batched seeds), then A/B on this box
mkdir -p gpurun_out/r3q
timeout 3000 python tools/refine_table.py tools/ab/tables/r3.json gpurun_out/r3q/tuned_r3_refined.json --cobatch 4 > gpurun_out/r3q/refine.log 2>&1; echo "refine rc=$?"; grep "refined" gpurun_out/r3q/refine.log
for i in 1 2; do
for tb in tools/ab/tables/quick.json tools/ab/tables/r3.json gpurun_out/r3q/tuned_r3_refined.json; do
  TMIX_TUNE_FILE=$tb timeout 400 python bench.py --kind lora --no-cpu-baseline --no-trajectory --no-video 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tb', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()})"
done; done
