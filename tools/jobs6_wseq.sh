#!/bin/bash
# round 6: would the chunk-major convolutions gain from weights STORED chunk-major (sequential 128-byte pieces per K-tile)?  Ablation build: the kernels walk the weight rows
# sequentially under the chunk-major A gather (wrong results, same traffic shape as a re-laid-out weight tensor would give).  Hot per shape, then the captured step.
out=gpurun_out/r6i; mkdir -p $out
for v in shipped wseq; do
  if [ $v = wseq ]; then export TMIX_LIB=tools/ab/wseq/libtmix_hip.so; else unset TMIX_LIB; fi
  echo "== $v"; python tools/convh_bench.py 10 2>/dev/null
  for r in 1 2; do python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1; done
done 2>&1 | tee $out/wseq.txt
