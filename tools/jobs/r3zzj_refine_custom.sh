# round 3, call ZZJ: the same refine for the Custom-Diffusion plans (shared-weight projections), on top of the table call ZZI produced; then lora / custom A/B against the previous table
mkdir -p gpurun_out/r3zzj
git show HEAD:tweediemix_amd/tuned_gfx950.json > gpurun_out/r3zzj/old.json 2>/dev/null || cp tools/ab/old_table.json gpurun_out/r3zzj/old.json
timeout 1500 python tools/refine_table.py tweediemix_amd/tuned_gfx950.json gpurun_out/r3zzj/refined.json --top 30 --reps 9 --kinds custom --cands 2,4,7,12,13,14,19,20,21 > gpurun_out/r3zzj/refine.log 2>&1; echo "refine rc=$?"
grep -E "refine \(" gpurun_out/r3zzj/refine.log | awk -F': ' '{split($2,a," "); if (a[1]!=a[3]) print}' | cut -c1-170; grep refined gpurun_out/r3zzj/refine.log
for i in 1 2; do
for kind in lora custom; do
for tb in gpurun_out/r3zzj/old.json tweediemix_amd/tuned_gfx950.json gpurun_out/r3zzj/refined.json; do
  TMIX_TUNE_FILE=$tb timeout 600 python bench.py --kind $kind --no-cpu-baseline --no-video --no-trajectory --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kind $tb', round(d['value'],2), round(d['ms_per_step'],3), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()})"
done; done; done
