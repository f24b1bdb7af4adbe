#!/bin/bash
# round 4: per-shape table of the fusion call with 4 seeds co-batched (B = 16, the images/s regime), bf16 and fp8
mkdir -p gpurun_out/r4za; rm -f gpurun_out/r4za/*
for dt in bf16 fp8; do
  timeout 600 python tools/step_shapes.py fusion --dtype $dt --seeds-per-gpu 4 > gpurun_out/r4za/$dt.out 2> gpurun_out/r4za/$dt.err
done
cat gpurun_out/r4za/bf16.out; grep -h "total=" gpurun_out/r4za/bf16.err | head -32; tail -5 gpurun_out/r4za/bf16.err; cat gpurun_out/r4za/fp8.out; grep -h "total=" gpurun_out/r4za/fp8.err | head -24
