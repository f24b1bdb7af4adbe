# round 3, call Z: tiling 21 = 128x160 with FOUR loader waves (one per SIMD): correctness, hot / cold timing, per-workgroup timeline
mkdir -p gpurun_out/r3z; rm -f gpurun_out/r3z/*
L=tools/gemm_lab/lab
timeout 300 $L check nocold 512,512,256,1,b 1024,1280,1280,1,br 300,264,128,1,b 2048,2560,1280,1,brs 4096,1280,320,1,br 520,640,64,1,br 520,640,128,1,br 520,640,192,1,br 520,640,256,1,br cfgs=21 reps=3 > gpurun_out/r3z/check.txt 2>&1
grep -c " ok" gpurun_out/r3z/check.txt; grep "WRONG\|rc " gpurun_out/r3z/check.txt | head
timeout 300 $L time 4096,1280,1280,1,br 4096,1280,5120,1,br 4096,3840,1280,1,b 2048,1280,1280,1,br 16384,640,640,1,br 16384,640,2560,1,br 4096,10240,1280,1,g cfgs=12,18,20,21 reps=30 > gpurun_out/r3z/time.txt 2>&1
cat gpurun_out/r3z/time.txt | cut -c1-160
echo "===== new" > gpurun_out/r3z/tl.txt
timeout 300 $L tl 4096,1280,1280,1,br 4096,1280,5120,1,br 2048,1280,1280,1,br 16384,640,2560,1,br cfgs=12,20,21 reps=20 >> gpurun_out/r3z/tl.txt 2>&1
python tools/tl_table.py gpurun_out/r3z/tl.txt
