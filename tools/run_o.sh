timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "groupnorm or linear_small" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_unet_gpu.py tests/test_sampler_gpu.py -m gpu -q -x 2>&1 | tail -3
