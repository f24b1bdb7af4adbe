#!/bin/bash
# round 4: tiling 22 (register-staging loaders) against 12 / 20 / 21 on the N = 1280 shapes -- correctness, hot / cold time, per-workgroup timeline
mkdir -p gpurun_out/r4b
L=tools/gemm_lab/lab
$L check 4096,1280,1280,1,br 4096,1280,5120,1,br 2048,1280,1280,1,br 4096,1280,320,1,b 4096,1280,128,1,b 4096,1280,64,1,b cfgs=12,21,22 reps=20 > gpurun_out/r4b/lab.log 2>&1
$L tl 4096,1280,1280,1,br 4096,1280,5120,1,br cfgs=12,20,21,22 reps=20 nocold >> gpurun_out/r4b/lab.log 2>&1
cat gpurun_out/r4b/lab.log | tail -60
