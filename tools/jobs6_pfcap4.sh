#!/bin/bash
# round 6: the weight-hint cap on the fp8 plan (routed e4m3 q/k/v weights are 19.7 MB: under the shipped 20 MB threshold) and on the video step
out=gpurun_out/r6z4; mkdir -p $out
{
for r in 1 2; do
  echo -n "fp8 lora fusion | shipped (over 20 cap 8): "; python tools/step_shapes.py fusion --kind lora --dtype fp8 2>/dev/null | tail -1
  echo -n "fp8 lora fusion | over 16 cap 8: "; TMIX_PF_CAP_OVER_MB=16 TMIX_PF_CAP_MB=8 python tools/step_shapes.py fusion --kind lora --dtype fp8 2>/dev/null | tail -1
  echo -n "fp8 lora fusion | over 16 cap 1: "; TMIX_PF_CAP_OVER_MB=16 TMIX_PF_CAP_MB=1 python tools/step_shapes.py fusion --kind lora --dtype fp8 2>/dev/null | tail -1
  echo -n "fp8 lora fusion | whole tensors: "; TMIX_PF_CAP_MB=0 python tools/step_shapes.py fusion --kind lora --dtype fp8 2>/dev/null | tail -1
  echo -n "video | shipped (whole tensors): "; python tools/video_one.py 2>/dev/null | tail -1
  echo -n "video | over 20 cap 8: "; TMIX_PF_CAP_OVER_MB=20 TMIX_PF_CAP_MB=8 python tools/video_one.py 2>/dev/null | tail -1
  echo -n "video | over 8 cap 4: "; TMIX_PF_CAP_OVER_MB=8 TMIX_PF_CAP_MB=4 python tools/video_one.py 2>/dev/null | tail -1
done
} 2>&1 | tee $out/pfcap4.txt
