#!/bin/bash
# round 6: the convolutions re-ranked in the captured step under the channel-chunk-major K order (the halo-patch kernel among the candidates), single-seed plans
out=gpurun_out/r6j; mkdir -p $out
python tools/refine_table.py tweediemix_amd/tuned_gfx950.json $out/refined.json --only-kind conv --cands 2,4,7,12,14,20,26 --top 60 --kinds lora 2>&1 | grep -E "refine |refined" | tee $out/refine_conv.txt
for v in shipped refined; do
  if [ $v = refined ]; then export TMIX_TUNE_FILE=$out/refined.json; else export TMIX_TUNE_FILE=tweediemix_amd/tuned_gfx950.json; fi
  for r in 1 2; do echo -n "$v: "; python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1; done
  echo -n "$v custom: "; python tools/step_shapes.py fusion --kind custom 2>/dev/null | tail -1
  echo -n "$v plain: "; python tools/step_shapes.py plain --kind lora 2>/dev/null | tail -1
done 2>&1 | tee -a $out/refine_conv.txt
