#!/bin/bash
# round 4: the hostile-statistics parity tests (tiny + SDXL-width UNet, bf16 and fp8 plans; folded LayerNorm against row means up to 1000 std)
mkdir -p gpurun_out/r4d
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_ops_gpu.py -q -s -m gpu -k "hostile or far_from_zero" > gpurun_out/r4d/hostile.log 2>&1
grep -E "hostile|fused LayerNorm|passed|failed|Error|assert" gpurun_out/r4d/hostile.log | tail -60
