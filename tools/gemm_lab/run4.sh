mkdir -p gpurun_out
L=tools/gemm_lab/lab
timeout 600 $L tl 4096,10240,1280,1,g 4096,3840,1280,1,b 4096,1280,1280,1,br 4096,1280,5120,1,br 16384,5120,640,1,g 16384,640,640,1,br 2048,1280,1280,1,br 4096,4096,4096 cfgs=2,4,7,9,14,16,17 reps=20 > gpurun_out/lab4_tl.txt 2>&1
cat gpurun_out/lab4_tl.txt
