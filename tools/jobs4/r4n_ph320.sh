#!/bin/bash
# round 4: tiling 22 (256 x 320, phase-offset loop) -- kernel tests for every tiling-parametrised GEMM test, then FF1 / q/k/v / FF2 hot and cold against 14 / 16
mkdir -p gpurun_out/r4n; rm -f gpurun_out/r4n/*
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "22" > gpurun_out/r4n/tests.log 2>&1; tail -3 gpurun_out/r4n/tests.log
L=tools/gemm_lab/lab
timeout 200 $L tl 4096,10240,1280,1,g 16384,5120,640,1,g 4096,3840,1280,1,b 2048,10240,1280,1,g cfgs=14,22,16 reps=20 > gpurun_out/r4n/lab.log 2>&1
grep -E "gemm|cfg|timeline" gpurun_out/r4n/lab.log | cut -c1-160
