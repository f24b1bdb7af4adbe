#!/bin/bash
# round 6: which of the refine's proposals hold in the captured step?  v0 = shipped + the 8-seed plans' entries, v1 = + tiling 23 on FF2 / unrouted to_out (row statistics) / the B = 2 64^2 FF2,
# v2 = v1 + routed q/k/v 16 -> 4, v3 = v1 + the 64^2 conv2 + shortcut(320) 26 -> 7; three interleaved rounds, LoRA and Custom-Diffusion fusion steps
out=gpurun_out/r6s; mkdir -p $out
for r in 1 2 3; do for v in v0 v1 v2 v3; do
  export TMIX_TUNE_FILE=tools/tables/r6_$v.json
  echo -n "$v lora: "; python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
  echo -n "$v custom: "; python tools/step_shapes.py fusion --kind custom 2>/dev/null | tail -1
done; done 2>&1 | tee $out/table_variants.txt
