# round 3, call F: pipelined loop incl. loader-wave tilings 19 / 20 and the rolling W buffer of tiling 14
mkdir -p gpurun_out/r3f; rm -f gpurun_out/r3f/*
L=tools/gemm_lab/lab
timeout 600 $L check nocold 512,512,256,1,b 1024,1280,1280,1,br 300,264,128,1,b 2048,2560,1280,1,brs 4096,1280,320,1,br 520,640,64,1,br cfgs=1,2,3,4,5,7,12,13,14,15,18,19,20 reps=3 > gpurun_out/r3f/check.txt 2>&1
grep -c " ok" gpurun_out/r3f/check.txt; grep "WRONG\|rc " gpurun_out/r3f/check.txt | head
for v in pipe0 new; do
  if [ $v = new ]; then LP=""; else LP=tools/ab/$v; fi
  echo "===== $v" >> gpurun_out/r3f/tl.txt
  LD_LIBRARY_PATH=$LP timeout 300 $L tl 4096,1280,1280,1,br 4096,1280,5120,1,br 2048,1280,1280,1,br 16384,640,2560,1,br cfgs=12,18,19,20 reps=20 >> gpurun_out/r3f/tl.txt 2>&1
  LD_LIBRARY_PATH=$LP timeout 300 $L tl 4096,10240,1280,1,g 16384,5120,640,1,g cfgs=14,4 reps=20 >> gpurun_out/r3f/tl.txt 2>&1
done
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -3
