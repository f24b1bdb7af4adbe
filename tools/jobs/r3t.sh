# round 3, call T: tiling 18 with the LDS-DMA issue of its two wave groups staggered (group 1 behind the hand-over)
mkdir -p gpurun_out/r3t; rm -f gpurun_out/r3t/*
L=tools/gemm_lab/lab
timeout 300 $L check nocold 512,512,256,1,b 1024,1280,1280,1,br 300,264,128,1,b 2048,2560,1280,1,brs 4096,1280,320,1,br 520,640,64,1,br 520,640,128,1,br 520,640,192,1,br 520,640,256,1,br cfgs=18 reps=3 2>&1 | grep -E "check|rc" | cut -c1-120
for i in 1 2 3; do timeout 100 $L check nocold 4096,1280,1280,1,br 4096,1280,5120,1,brs cfgs=18 reps=5 2>&1 | grep -E "check|rc" | cut -c1-120; done
echo "===== new" > gpurun_out/r3t/tl.txt
timeout 300 $L tl 4096,1280,1280,1,br 4096,1280,5120,1,br 2048,1280,1280,1,br 16384,640,2560,1,br cfgs=12,18,20 reps=20 >> gpurun_out/r3t/tl.txt 2>&1
python tools/tl_table.py gpurun_out/r3t/tl.txt
