# round 3, call V: do two launch chains gain when every workgroup leaves room for a second one on its CU (LDS <= 80 KB tilings only)?
for t in 7 1; do for s in 1 2; do
  TMIX_TUNE_FILE=tweediemix_amd/tuned_gfx950.json TMIX_FORCE_TILE=$t timeout 400 python bench.py --kind lora --streams $s --no-cpu-baseline --no-trajectory --no-video 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force tile $t streams $s:', round(d['value'],2), round(d['ms_per_step'],2), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()})"
done; done
timeout 400 python bench.py --kind lora --streams 1 --no-cpu-baseline --no-trajectory --no-video 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default:', round(d['value'],2), round(d['ms_per_step'],2))"
