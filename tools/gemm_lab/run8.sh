mkdir -p gpurun_out
L=tools/gemm_lab/lab
timeout 600 $L check nocold 512,512,256,1,b 1024,1280,1280,1,br 300,264,128,1,b 2048,2560,1280,1,brs cfgs=12,18 reps=3 2>&1 | grep -E "check|gemm"
timeout 600 $L tl 4096,1280,1280,1,br 4096,1280,5120,1,br 4096,1280,1280,1,bs 16384,640,640,1,br 16384,640,2560,1,br 2048,1280,1280,1,br cfgs=7,12,18,9,17 reps=20 2>&1 | grep -v "^$"
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "gemm or wide" 2>&1 | tail -3
