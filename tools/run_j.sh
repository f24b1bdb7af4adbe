mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py -m gpu -q 2>&1 | tail -6
