#!/bin/bash
# round 6: does the graph-timed refine pick tiling 23 (128x160 over 2x2 waves, gemm_w22.hip) anywhere under the partition-mask workload?
out=gpurun_out/r6n; mkdir -p $out
python tools/refine_table.py tweediemix_amd/tuned_gfx950.json $out/refined23.json --only-kind gemm --cands 23 --top 40 --kinds lora 2>&1 | grep -E "refine |refined" | tee $out/refine23.txt
for r in 1 2; do for v in shipped refined23; do
  if [ $v = refined23 ]; then export TMIX_TUNE_FILE=$out/refined23.json; else export TMIX_TUNE_FILE=tweediemix_amd/tuned_gfx950.json; fi
  echo -n "$v: "; python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
done; done 2>&1 | tee -a $out/refine23.txt
