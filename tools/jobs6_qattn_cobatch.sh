#!/bin/bash
# round 6: the one-launch attn2 (tmix_gemm_q_cross_attn) against the to_q GEMM + attention pair on the CO-BATCHED plans (8 seeds: B = 32 fusion / start, B = 16 plain), same box
out=gpurun_out/r6h; mkdir -p $out
for v in fused pair; do
  if [ $v = pair ]; then export TMIX_NO_QATTN=1; else unset TMIX_NO_QATTN; fi
  for k in fusion plain; do
    echo -n "$v $k: "; python tools/step_shapes.py $k --seeds-per-gpu 8 --kind lora 2>/dev/null | tail -1
  done
  for s in 2 4; do echo -n "$v fusion seeds=$s: "; python tools/step_shapes.py fusion --seeds-per-gpu $s --kind lora 2>/dev/null | tail -1; done
done 2>&1 | tee $out/qattn_cobatch.txt
