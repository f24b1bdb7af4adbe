#!/bin/bash
# round 4: graph-timed refine of the single-seed table with this round's kernels (new candidates: 21 with four loaders, 22, 20 for shortcut convs), lora plans only
mkdir -p gpurun_out/r4t
timeout 2400 python tools/refine_table.py tweediemix_amd/tuned_gfx950.json gpurun_out/r4t/refined.json --top 30 --kinds lora > gpurun_out/r4t/refine.log 2>&1
grep -E "refined|->|wrote" gpurun_out/r4t/refine.log | tail -30
for t in tweediemix_amd/tuned_gfx950.json gpurun_out/r4t/refined.json; do
  TMIX_TUNE_FILE=$t timeout 600 python bench.py --kind lora --no-trajectory --no-cpu-baseline --no-video --steps 50 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['value'], d['ms_per_step'])"
done
