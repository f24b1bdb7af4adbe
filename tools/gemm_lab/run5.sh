mkdir -p gpurun_out
L=tools/gemm_lab/lab
timeout 600 $L check nocold 512,512,256,1,b 1024,1280,1280,1,br 300,264,128,1,b 2048,2560,1280,1,brs 1000,640,320,1,br cfgs=1,2,3,4,5,6,7,9,12,13,14,15,16,17 reps=3 > gpurun_out/lab5_check.txt 2>&1
timeout 600 $L tl 4096,10240,1280,1,g 4096,3840,1280,1,b 4096,1280,1280,1,br 4096,1280,5120,1,br 16384,5120,640,1,g 16384,640,640,1,br 2048,1280,1280,1,br 1024,3840,1280,4,bt cfgs=2,4,7,9,14,16,17 reps=20 > gpurun_out/lab5_tl.txt 2>&1
TMIX_NARROW_EPILOGUE=1 timeout 600 $L nocold 4096,3840,1280,1,b 4096,1280,1280,1,br 1024,3840,1280,4,bt cfgs=2,9,16 reps=20 > gpurun_out/lab5_narrow.txt 2>&1
grep -c "ok$" gpurun_out/lab5_check.txt; grep -v "ok$" gpurun_out/lab5_check.txt; cat gpurun_out/lab5_tl.txt gpurun_out/lab5_narrow.txt
