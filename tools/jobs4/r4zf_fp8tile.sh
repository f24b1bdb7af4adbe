#!/bin/bash
# round 4: size-aware fp8 tiling + periodic weight sets: B = 16 fp8 call per shape, the fp8 headline leg, and the UNet / sampler tests
mkdir -p gpurun_out/r4zf; rm -f gpurun_out/r4zf/*
timeout 600 python tools/step_shapes.py fusion --dtype fp8 --seeds-per-gpu 4 > gpurun_out/r4zf/fp8_b16.out 2> gpurun_out/r4zf/fp8_b16.err
timeout 600 python tools/step_shapes.py fusion --dtype fp8 > gpurun_out/r4zf/fp8_b4.out 2> gpurun_out/r4zf/fp8_b4.err
cat gpurun_out/r4zf/fp8_b16.out gpurun_out/r4zf/fp8_b4.out
timeout 2400 python -m pytest tests/test_unet_gpu.py tests/test_sampler_gpu.py -x -q 2>&1 | tail -5 | tee gpurun_out/r4zf/tests.log
