#!/bin/bash
# round 4: valid per-shape table of the video step (every instrumented launch has its meta entry now), one chain and two; then a short bench sanity (slot accounting assert)
mkdir -p gpurun_out/r4zh; rm -f gpurun_out/r4zh/*
for s in 1 2; do timeout 900 python tools/video_step_shapes.py $s > gpurun_out/r4zh/s$s.out 2> gpurun_out/r4zh/s$s.err; done
cat gpurun_out/r4zh/s1.out; tail -3 gpurun_out/r4zh/s1.err; head -1 gpurun_out/r4zh/s2.out
timeout 900 python bench.py --kind lora --steps 10 --warmup 3 --no-trajectory --no-cpu-baseline > gpurun_out/r4zh/bench.json 2> gpurun_out/r4zh/bench.err; tail -c 600 gpurun_out/r4zh/bench.json; tail -3 gpurun_out/r4zh/bench.err
