timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "fp8" 2>&1 | tail -8
tools/gemm_lab/lab 4096,1280,5120,1,brf 4096,1280,5120,1,rf 4096,10240,1280,1,gf cfgs=16,17 reps=20 nocold 2>&1 | grep cfg
