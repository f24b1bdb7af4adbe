#!/bin/bash
# round 4: per-shape table of the B = 2 call (plain CFG steps of the trajectory), bf16 and fp8
mkdir -p gpurun_out/r4z; rm -f gpurun_out/r4z/*
for dt in bf16 fp8; do
  timeout 600 python tools/step_shapes.py plain --dtype $dt > gpurun_out/r4z/$dt.out 2> gpurun_out/r4z/$dt.err
done
cat gpurun_out/r4z/bf16.out; grep -h "total=" gpurun_out/r4z/bf16.err | head -40; tail -5 gpurun_out/r4z/bf16.err
