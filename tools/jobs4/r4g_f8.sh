#!/bin/bash
# round 4: e4m3 operands in the lock-step loops (tilings 12 / 21 / 14 under tmix_gemm_fp8): parity tests, then hot / cold times against the phase-offset tilings and bf16
mkdir -p gpurun_out/r4g; rm -f gpurun_out/r4g/*
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "fp8" > gpurun_out/r4g/tests.log 2>&1; tail -5 gpurun_out/r4g/tests.log
L=tools/gemm_lab/lab
timeout 200 $L time 4096,1280,1280,1,brf 4096,1280,5120,1,brf 4096,3840,1280,1,bf 4096,10240,1280,1,gf 16384,640,640,1,brf cfgs=12,21,14,16,17 reps=20 > gpurun_out/r4g/lab_f8.log 2>&1
timeout 200 $L time 4096,1280,1280,1,br 4096,1280,5120,1,br 4096,10240,1280,1,g cfgs=21,14 reps=20 > gpurun_out/r4g/lab_bf16.log 2>&1
grep -E "gemm|cfg" gpurun_out/r4g/lab_f8.log gpurun_out/r4g/lab_bf16.log | cut -c1-150
