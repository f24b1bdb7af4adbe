#!/bin/bash
# round 4: ablations of the 256 x 320 loop (tiling 14, FF1 + GEGLU) and of the phase-offset 256 x 256 loop on the q/k/v shape
mkdir -p gpurun_out/r4f; rm -f gpurun_out/r4f/*
L=tools/gemm_lab/lab
for v in new abl2 abl4 abl6; do
  if [ $v = new ]; then LP=""; else LP=tools/ab/$v; fi
  echo "===== $v" >> gpurun_out/r4f/tl.txt
  LD_LIBRARY_PATH=$LP timeout 120 $L tl 4096,10240,1280,1,g cfgs=14 reps=20 nocold >> gpurun_out/r4f/tl.txt 2>&1
  LD_LIBRARY_PATH=$LP timeout 120 $L tl 2048,10240,1280,1,g cfgs=14 reps=20 nocold >> gpurun_out/r4f/tl.txt 2>&1
done
grep -E "=====|gemm|timeline" gpurun_out/r4f/tl.txt
