# round 3, call C: the software-pipelined K loop (PIPE) against the round-2 loop (tools/ab/pipe0), same box: correctness on odd
# shapes for every plain tiling, then timelines of the step's main shapes, then the GEMM/conv unit tests
mkdir -p gpurun_out/r3c
L=tools/gemm_lab/lab
timeout 600 $L check nocold 512,512,256,1,b 1024,1280,1280,1,br 300,264,128,1,b 2048,2560,1280,1,brs 4096,1280,320,1,br cfgs=1,2,3,4,5,6,7,12,13,14,15,18 reps=3 > gpurun_out/r3c/check.txt 2>&1
grep -c " ok" gpurun_out/r3c/check.txt; grep -c "WRONG\|rc " gpurun_out/r3c/check.txt
for v in pipe0 new; do
  if [ $v = new ]; then LP=""; else LP=tools/ab/$v; fi
  echo "===== $v" >> gpurun_out/r3c/tl.txt
  LD_LIBRARY_PATH=$LP timeout 300 $L tl 4096,1280,1280,1,br 4096,1280,5120,1,br 2048,1280,1280,1,br 16384,640,640,1,br 16384,640,2560,1,br cfgs=7,12,13,18,1,2 reps=20 >> gpurun_out/r3c/tl.txt 2>&1
  LD_LIBRARY_PATH=$LP timeout 300 $L tl 4096,3840,1280,1,b cfgs=4,6,16 reps=20 >> gpurun_out/r3c/tl.txt 2>&1
  LD_LIBRARY_PATH=$LP timeout 300 $L tl 4096,10240,1280,1,g 16384,5120,640,1,g cfgs=14,4,16,6 reps=20 >> gpurun_out/r3c/tl.txt 2>&1
done
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm or conv or wide" 2>&1 | tail -3
