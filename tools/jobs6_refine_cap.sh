#!/bin/bash
# round 6: graph-timed refine of the single-seed plans under the capped weight hints (UNetPlan._pf_cap), then shipped vs refined table
out=gpurun_out/r6q; mkdir -p $out
python tools/refine_table.py tweediemix_amd/tuned_gfx950.json $out/refined.json --cands 1,2,3,4,5,7,12,13,14,15,16,17,18,19,20,21,22,23,26 --top 36 --kinds lora,custom 2>&1 | grep -E "refine |refined" > $out/refine_single.txt; grep refined $out/refine_single.txt
for r in 1 2; do for v in shipped refined uncapped; do
  unset TMIX_PF_CAP_MB; export TMIX_TUNE_FILE=tweediemix_amd/tuned_gfx950.json
  if [ $v = refined ]; then export TMIX_TUNE_FILE=$out/refined.json; fi
  if [ $v = uncapped ]; then export TMIX_PF_CAP_MB=0; fi
  echo -n "$v: "; python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
  echo -n "$v custom: "; python tools/step_shapes.py fusion --kind custom 2>/dev/null | tail -1
  echo -n "$v plain: "; python tools/step_shapes.py plain --kind lora 2>/dev/null | tail -1
  echo -n "$v start: "; python tools/step_shapes.py start --kind lora 2>/dev/null | tail -1
done; done 2>&1 | tee $out/refine_ab.txt
