#!/bin/bash
# round 4: convolution on e4m3 operands -- kernel tests (GroupNorm e4m3 output, conv fp8 vs the conv of the dequantised operands), then times against bf16
mkdir -p gpurun_out/r4o; rm -f gpurun_out/r4o/*
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "e4m3_with_row_major or conv3x3_fp8" > gpurun_out/r4o/tests.log 2>&1; tail -15 gpurun_out/r4o/tests.log
timeout 300 python tools/conv_abl.py > gpurun_out/r4o/conv_time.log 2>&1; cat gpurun_out/r4o/conv_time.log
