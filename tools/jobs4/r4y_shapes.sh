#!/bin/bash
# round 4: per-shape table of the headline step (bf16, then fp8): in-situ device-clock stamps summed per launch shape + kernel boundaries by class pair
mkdir -p gpurun_out/r4y; rm -f gpurun_out/r4y/*
for dt in bf16 fp8; do
  TMIX_BENCH_SHAPES=1 timeout 600 python bench.py --kind lora --dtype $dt --steps 20 --warmup 5 --no-trajectory --no-video --no-cpu-baseline > gpurun_out/r4y/$dt.json 2> gpurun_out/r4y/$dt.err
done
grep -h "total=" gpurun_out/r4y/bf16.err | head -50
