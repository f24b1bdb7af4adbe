mkdir -p gpurun_out
L=tools/gemm_lab/lab
timeout 300 $L check 512,512,256,1,b 1024,1280,1280,1,br 300,260,128,1,b 2048,2560,1280,1,br cfgs=2,16,17 reps=5 nocold > gpurun_out/lab2_check.txt 2>&1
timeout 600 $L 4096,10240,1280,1,g 4096,3840,1280,1,b 4096,1280,1280,1,br 4096,1280,5120,1,br 16384,5120,640,1,g 16384,640,640,1,br 8192,8192,8192 4096,4096,4096 cfgs=4,9,11,14,16,17 reps=20 > gpurun_out/lab2_time.txt 2>&1
cat gpurun_out/lab2_check.txt; cat gpurun_out/lab2_time.txt
