#!/bin/bash
# round 4: where does a convolution K-tile's time go?  shipped library and the ablation builds (2 no MFMAs, 4 no LDS-DMA in the loop [non-loader tilings], 6 neither)
mkdir -p gpurun_out/r4l; rm -f gpurun_out/r4l/*
for v in shipped abl2 abl4 abl6; do
  if [ $v = shipped ]; then unset TMIX_LIB; else export TMIX_LIB=tools/ab/$v/libtmix_hip.so; fi
  timeout 200 python tools/conv_abl.py >> gpurun_out/r4l/conv_abl.log 2>&1
done
cat gpurun_out/r4l/conv_abl.log
