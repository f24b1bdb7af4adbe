#!/bin/bash
# round 4: per-shape table of the video step (config #5), one chain and two chains
mkdir -p gpurun_out/r4zc; rm -f gpurun_out/r4zc/*
for s in 2 1; do timeout 900 python tools/video_step_shapes.py $s > gpurun_out/r4zc/s$s.out 2> gpurun_out/r4zc/s$s.err; done
cat gpurun_out/r4zc/s2.out; tail -3 gpurun_out/r4zc/s2.err; head -3 gpurun_out/r4zc/s1.out
