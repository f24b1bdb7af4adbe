# round 3, call ZG: straight-line epilogue also on the tilings without residual registers (256x256, 256x320) when the launch has no residual
mkdir -p gpurun_out/r3zg; rm -f gpurun_out/r3zg/*
L=tools/gemm_lab/lab
timeout 600 $L check nocold 512,512,256,1,b 512,512,256,1 1024,1280,1280,1,br 1024,1280,1280,1,r 300,264,128,1,b 300,264,128,1,br 2048,2560,1280,1,brs 2048,2560,1280,1,s 2048,2560,1280,1,bs 520,648,64,1,brs 1024,3840,1280,4,bt cfgs=4,14,16,5 reps=3 > gpurun_out/r3zg/check.txt 2>&1
grep -c " ok" gpurun_out/r3zg/check.txt; grep "WRONG\|rc " gpurun_out/r3zg/check.txt | head
echo "===== new" >> gpurun_out/r3zg/tl.txt
timeout 300 $L tl 1024,3840,1280,4,bt 4096,1920,640,1,b cfgs=16,4 reps=20 >> gpurun_out/r3zg/tl.txt 2>&1
python tools/tl_table.py gpurun_out/r3zg/tl.txt
