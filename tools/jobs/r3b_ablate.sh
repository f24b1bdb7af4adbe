# round 3, call B: what bounds the K loop of this path's GEMM launches?  The same launches with (1) every workgroup staging
# tile (0,0) -> operands L2-hot, (2) no MFMAs, (4) no LDS-DMA inside the loop, (8) no epilogue; per-workgroup timelines.
mkdir -p gpurun_out/r3b
L=tools/gemm_lab/lab
for v in base abl1 abl2 abl4 abl8 abl3 abl6; do
  if [ $v = base ]; then LP=""; else LP=tools/ab/$v; fi
  echo "===== $v" >> gpurun_out/r3b/abl.txt
  LD_LIBRARY_PATH=$LP timeout 300 $L tl 4096,1280,1280,1,br 4096,1280,5120,1,br cfgs=12,18,17 reps=20 >> gpurun_out/r3b/abl.txt 2>&1
  LD_LIBRARY_PATH=$LP timeout 300 $L tl 4096,3840,1280,1,b cfgs=4,16 reps=20 >> gpurun_out/r3b/abl.txt 2>&1
  LD_LIBRARY_PATH=$LP timeout 300 $L tl 4096,10240,1280,1,g cfgs=14,16 reps=20 >> gpurun_out/r3b/abl.txt 2>&1
done
grep -c timeline gpurun_out/r3b/abl.txt
