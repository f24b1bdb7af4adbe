#!/bin/bash
# round 6: graph-timed refine of the shipped table under this round's kernels -- single-seed plans (every candidate incl. 23 and 26), then the heaviest shapes of the 8-seed plans
out=gpurun_out/r6r; mkdir -p $out
python tools/refine_table.py tweediemix_amd/tuned_gfx950.json $out/refined.json --cands 1,2,3,4,5,7,12,13,14,15,16,17,18,19,20,21,22,23,26 --top 36 --kinds lora,custom 2>&1 | grep -E "refine |refined" > $out/refine_single.txt; grep refined $out/refine_single.txt
python tools/refine_table.py $out/refined.json $out/refined_co.json --only-cobatch --cobatch 8 --cands 2,4,14,16,17,22 --top 24 --kinds lora 2>&1 | grep -E "refine |refined" > $out/refine_cobatch.txt; grep refined $out/refine_cobatch.txt
for r in 1 2; do for v in shipped refined; do
  if [ $v = refined ]; then export TMIX_TUNE_FILE=$out/refined_co.json; else export TMIX_TUNE_FILE=tweediemix_amd/tuned_gfx950.json; fi
  echo -n "$v: "; python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
  echo -n "$v custom: "; python tools/step_shapes.py fusion --kind custom 2>/dev/null | tail -1
  echo -n "$v plain: "; python tools/step_shapes.py plain --kind lora 2>/dev/null | tail -1
  echo -n "$v cobatch8 fusion: "; python tools/step_shapes.py fusion --kind lora --seeds-per-gpu 8 2>/dev/null | tail -1
  echo -n "$v cobatch8 plain: "; python tools/step_shapes.py plain --kind lora --seeds-per-gpu 8 2>/dev/null | tail -1
done; done 2>&1 | tee $out/refine_ab.txt
