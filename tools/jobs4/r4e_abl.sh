#!/bin/bash
# round 4: ablations of the loader-wave loop (tiling 21) on the hot FF2 / cube launches: 2 = no MFMAs, 4 = no LDS-DMA inside the loop, 6 = neither
mkdir -p gpurun_out/r4e; rm -f gpurun_out/r4e/*
L=tools/gemm_lab/lab
for v in new abl2 abl4 abl6; do
  if [ $v = new ]; then LP=""; else LP=tools/ab/$v; fi
  echo "===== $v" >> gpurun_out/r4e/tl.txt
  LD_LIBRARY_PATH=$LP timeout 120 $L tl 4096,1280,5120,1,br 4096,1280,1280,1,br cfgs=12,20,21 reps=20 nocold >> gpurun_out/r4e/tl.txt 2>&1
done
grep -E "=====|gemm|cfg|timeline" gpurun_out/r4e/tl.txt
