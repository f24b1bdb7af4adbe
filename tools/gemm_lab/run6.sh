mkdir -p gpurun_out
L=tools/gemm_lab/lab
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "fp8 or gemm or conv or tweedie or prof or prologue" > gpurun_out/pytest_ops.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/pytest_ops.log
timeout 600 $L check nocold 1024,3840,1280,4,bt 512,1536,256,2,bt cfgs=2,4,7,14,16,17 reps=3 > gpurun_out/lab6_check.txt 2>&1
timeout 600 $L tl 4096,10240,1280,1,g 4096,10240,1280,1,gf 4096,1280,5120,1,br 4096,1280,5120,1,brf 1024,3840,1280,4,bt 1024,3840,1280,4,btf 4096,1280,1280,1,br 4096,1280,1280,1,brf 16384,5120,640,1,g 16384,5120,640,1,gf 8192,8192,8192 8192,8192,8192,1,f cfgs=2,14,16,17 reps=20 > gpurun_out/lab6_tl.txt 2>&1
cat gpurun_out/lab6_check.txt | tail -20; cat gpurun_out/lab6_tl.txt
