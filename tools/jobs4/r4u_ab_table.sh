#!/bin/bash
# round 4: does the graph-timed refine of r4t reproduce on another box?  shipped table vs refined, twice each, interleaved
mkdir -p gpurun_out/r4u
for i in 1 2; do
for t in tweediemix_amd/tuned_gfx950.json tools/ab/r4t_refined.json; do
  TMIX_TUNE_FILE=$t timeout 600 python bench.py --kind lora --no-trajectory --no-cpu-baseline --no-video --steps 50 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['value'], d['ms_per_step'], {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()})"
done
done
