#!/bin/bash
mkdir -p gpurun_out/r4m
bash tools/jobs4/r4l_convabl.sh > gpurun_out/r4m/convabl.out 2>&1
timeout 1500 python -m pytest tests/test_sampler_gpu.py tests/test_text_gpu.py tests/test_unet_gpu.py tests/test_vae_gpu.py tests/test_video_gpu.py tests/test_i2vgen_gpu.py -q -m gpu > gpurun_out/r4m/tests.log 2>&1; tail -4 gpurun_out/r4m/tests.log
