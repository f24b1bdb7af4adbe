# round 3, call ZD: straight-line staged epilogue (bias / residual from registers filled under the last K-tile): correctness of every tiling, timeline
mkdir -p gpurun_out/r3zd; rm -f gpurun_out/r3zd/*
L=tools/gemm_lab/lab
timeout 600 $L check nocold 512,512,256,1,b 512,512,256,1 1024,1280,1280,1,br 1024,1280,1280,1,r 300,264,128,1,b 300,264,128,1,br 2048,2560,1280,1,brs 2048,2560,1280,1,s 4096,1280,320,1,br 520,640,64,1,br 520,648,64,1,brs 1024,2560,640,1,g cfgs=1,2,3,4,5,7,12,13,14,15,16,17,18,19,20,21 reps=3 > gpurun_out/r3zd/check.txt 2>&1
grep -c " ok" gpurun_out/r3zd/check.txt; grep "WRONG\|rc " gpurun_out/r3zd/check.txt | head
echo "===== new" >> gpurun_out/r3zd/tl.txt
timeout 300 $L tl 4096,1280,1280,1,br 4096,1280,1280,1,b 4096,1280,1280,1,brs 4096,1280,5120,1,br 4096,3840,1280,1,b 2048,1280,1280,1,br 16384,640,640,1,br cfgs=12,21 reps=20 >> gpurun_out/r3zd/tl.txt 2>&1
timeout 300 $L tl 4096,3840,1280,1,b 4096,1280,1280,1,br cfgs=16,17,2 reps=20 >> gpurun_out/r3zd/tl.txt 2>&1
python tools/tl_table.py gpurun_out/r3zd/tl.txt
